#!/usr/bin/env python3
"""Known-answer vectors for the build-defined random numbers of the renderer: the 32-bit mixing hash, the jittered
sample positions and the occlusion-ray directions.

Everything here is computed from the DEFINITIONS (written out below) with Python integers and numpy float32 scalars;
nothing is imported from oracle/ or renderer_amd/ and no generated table (.inc) is read, so an error in the C++
restatements, in the kernels or in a shared table cannot hide behind itself.  Writes tests/golden/kats.json.

    python tests/golden/make_kats.py

Definitions
  hash32(x)            x ^= x >> 16; x *= 0x7feb352d; x ^= x >> 15; x *= 0x846ca68b; x ^= x >> 16   (mod 2^32)
  sample 0             (0.5, 0.5): the pixel centre
  sample s > 0         k = hash32(seed ^ 0x9E3779B9); a = hash32(pixel + k); b = hash32(a ^ (s * 0x85EBCA6B));
                       (sx, sy) = ((hash32(b) >> 8) / 2^24, (hash32(b ^ 0x68E31DA4) >> 8) / 2^24)
  occlusion ray j      of `rays` about unit normal n, pixel hash hp = hash32(pixel ^ hash32(seed ^ 0xA0761D65)):
                       basis (Duff et al. 2017): sg = copysign(1, n.z); a = -1 / (sg + n.z); b = n.x n.y a;
                       T = (1 + sg n.x n.x a, sg b, -sg n.x), B = (b, sg + n.y n.y a, -n.y);
                       angle index i = ((j * 256) // rays + (hp & 255)) & 255, phi_i = 2 pi (i + 0.5) / 256;
                       u1 = (hash32(hp + j * 0x9E3779B1 + 1) >> 8) / 2^24; r = sqrt(u1);
                       w = (r cos phi_i) T + (r sin phi_i) B + sqrt(1 - u1) n
                       (all in float32, products and sums in the order written; cos / sin rounded once to float32)
"""
import json
import math
import os

import numpy as np

F = np.float32
M = 0xFFFFFFFF


def hash32(x):
    x &= M
    x ^= x >> 16
    x = (x * 0x7FEB352D) & M
    x ^= x >> 15
    x = (x * 0x846CA68B) & M
    x ^= x >> 16
    return x


def sample_offset(pixel, s, seed):
    if s == 0:
        return F(0.5), F(0.5)
    k = hash32(seed ^ 0x9E3779B9)
    a = hash32((pixel + k) & M)
    b = hash32(a ^ ((s * 0x85EBCA6B) & M))
    return F(hash32(b) >> 8) * F(2.0 ** -24), F(hash32(b ^ 0x68E31DA4) >> 8) * F(2.0 ** -24)


def angle_table():
    """(cos, sin) of phi_i rounded to float32.  math.cos / math.sin are accurate to well under an ulp of a double;
    an entry would round differently from the exact value only if it lay within 2^-53 of a float32 rounding boundary
    (the arguments are reduced by symmetry to [0, pi/4] first, where the libm kernels are at their best)."""
    out = []
    for i in range(256):
        # phi = 2 pi (i + 0.5) / 256 = pi (2 i + 1) / 256: reduce the odd multiple of pi/256 to the first octant
        m = (2 * i + 1) % 512          # phi = m pi / 256, m odd in [1, 511]
        q, r = divmod(m, 128)          # quadrant q (of pi/2), remainder r pi / 256 in (0, pi/2)
        if r <= 64:
            c, s = math.cos(r * math.pi / 256), math.sin(r * math.pi / 256)
        else:                          # cos x = sin(pi/2 - x)
            c, s = math.sin((128 - r) * math.pi / 256), math.cos((128 - r) * math.pi / 256)
        c, s = [(c, s), (-s, c), (-c, -s), (s, -c)][q]
        out.append((F(c), F(s)))
    return out


def ao_directions(e1, e2, pixel, seed, rays, table):
    e1 = [F(v) for v in e1]
    e2 = [F(v) for v in e2]
    n = [F(e1[1] * e2[2]) - F(e1[2] * e2[1]), F(e1[2] * e2[0]) - F(e1[0] * e2[2]), F(e1[0] * e2[1]) - F(e1[1] * e2[0])]
    nn = F(F(F(n[0] * n[0]) + F(n[1] * n[1])) + F(n[2] * n[2]))
    inv = F(F(1.0) / F(np.sqrt(nn)))
    n = [F(v * inv) for v in n]
    sg = F(math.copysign(1.0, float(n[2])))
    a = F(F(-1.0) / F(sg + n[2]))
    b = F(F(n[0] * n[1]) * a)
    T = [F(F(1.0) + F(F(F(sg * n[0]) * n[0]) * a)), F(sg * b), F(-F(sg * n[0]))]
    B = [b, F(sg + F(F(n[1] * n[1]) * a)), F(-n[1])]
    hp = hash32(pixel ^ hash32(seed ^ 0xA0761D65))
    out = []
    for j in range(rays):
        i = ((j * 256) // rays + (hp & 255)) & 255
        u1 = F(hash32((hp + j * 0x9E3779B1 + 1) & M) >> 8) * F(2.0 ** -24)
        r = F(np.sqrt(u1))
        x, y = F(r * table[i][0]), F(r * table[i][1])
        z = F(np.sqrt(F(F(1.0) - u1)))
        out.append([float(F(F(F(x * T[c]) + F(y * B[c])) + F(z * n[c]))) for c in range(3)])
    return out


def main():
    rng = np.random.default_rng(20261111)
    doc = {"hash32": [], "sample_offset": [], "angle_table_hex": [], "ao_directions": []}
    for x in [0, 1, 2, 0xFFFFFFFF, 0x80000000, 0x9E3779B9, 12345] + [int(v) for v in rng.integers(0, 2 ** 32, 25)]:
        doc["hash32"].append([x, hash32(x)])
    for pixel, s, seed in [(0, 0, 0), (0, 1, 0), (1920 * 540 + 960, 15, 0), (5, 3, 1), (5, 3, 2), (2 ** 31 - 1, 4095, 0xFFFFFFFF)] + \
            [(int(rng.integers(0, 2 ** 31)), int(rng.integers(1, 64)), int(rng.integers(0, 2 ** 32))) for _ in range(26)]:
        sx, sy = sample_offset(pixel, s, seed)
        doc["sample_offset"].append([pixel, s, seed, float(sx).hex(), float(sy).hex()])
    table = angle_table()
    doc["angle_table_hex"] = [[float(c).hex(), float(s).hex()] for c, s in table]
    cases = [([0.01, 0, 0], [0, 0.01, 0], 7, 3, 4), ([0, 0.02, 0], [0.01, 0, 0], 1000, 0, 16),  # normals +z and -z
             ([0.003, 0.001, -0.002], [-0.001, 0.004, 0.0005], 1920 * 400 + 1000, 0, 64)]
    for _ in range(5):
        cases.append(([float(F(v)) for v in rng.normal(0, 0.01, 3)], [float(F(v)) for v in rng.normal(0, 0.01, 3)],
                      int(rng.integers(0, 2 ** 21)), int(rng.integers(0, 2 ** 32)), int(rng.choice([7, 16, 64, 100]))))
    for e1, e2, pixel, seed, rays in cases:
        w = ao_directions(e1, e2, pixel, seed, rays, table)
        doc["ao_directions"].append({"e1": [float(F(v)).hex() for v in e1], "e2": [float(F(v)).hex() for v in e2], "pixel": pixel,
                                     "seed": seed, "rays": rays, "w_hex": [[float(F(c)).hex() for c in row] for row in w]})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kats.json")
    with open(path, "w") as f:
        json.dump(doc, f, indent=0)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
