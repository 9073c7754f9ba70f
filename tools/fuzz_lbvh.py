"""Random scenes through the GPU LBVH build against the oracle's sequential restatement: sorted keys, order, hierarchy,
parents and every box, bit for bit.  Sizes around the sort tile (2048 keys) and the refit chunk (512 leaves) come up
often, sizes of several spans of 256 chunks now and then; some scenes hold exact duplicates (equal Morton keys: the index tie-break), one triangle up to 200 times (runs
of equal keys on both sides of the short sort's limit) or are flat (one axis without extent).

    python tools/fuzz_lbvh.py [--scenes 300] [--seed 1] [--max-triangles 60000]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import objfmt  # noqa: E402
import renderer_amd as R  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-triangles", type=int, default=60000)
    a = ap.parse_args()
    O.build()
    rng = np.random.default_rng(a.seed)
    ctx = R.Context(0)
    total = 0
    for i in range(a.scenes):
        pick = rng.random()
        if pick < 0.3:  # around a multiple of the sort tile or the refit chunk
            n = max(2, int(rng.choice([512, 1024, 2048, 4096, 8192])) * int(rng.integers(1, 6)) + int(rng.integers(-2, 3)))
        elif pick < 0.5:
            n = int(rng.integers(2, 600))
        elif pick < 0.56:  # more than one span of 256 refit chunks: the climb above the chunks takes two launches
            n = int(rng.integers(131073, 700000))
        else:
            n = int(rng.integers(2, a.max_triangles))
        xyz, faces = objfmt.random_soup(n, seed=int(rng.integers(0, 1 << 30)), size=float(rng.choice([0.001, 0.01, 0.05])))
        if rng.random() < 0.3 and n > 20:  # exact duplicates of a stretch of triangles, elsewhere in the order
            m = int(rng.integers(1, n // 2))
            src = int(rng.integers(0, n - m))
            dst = int(rng.integers(0, n - m))
            xyz = xyz.reshape(n, 3, 3)
            xyz[dst:dst + m] = xyz[src:src + m].copy()
            xyz = xyz.reshape(-1, 3)
        if rng.random() < 0.25 and n > 300:  # one triangle many times over: a long run of equal keys
            xyz = xyz.reshape(n, 3, 3)
            m = int(rng.integers(2, 200))
            xyz[rng.choice(n, m, replace=False)] = xyz[int(rng.integers(0, n))].copy()
            xyz = xyz.reshape(-1, 3)
        if rng.random() < 0.15:  # a flat scene
            xyz[:, int(rng.integers(0, 3))] = 0.05
        xyzw = np.zeros((len(xyz), 4), np.float32)
        xyzw[:, :3] = xyz
        idx = faces.reshape(-1).astype(np.uint32)
        scene = R.Scene(ctx, R.Mesh.from_arrays(xyzw, idx))
        got, want = scene.read_lbvh(), O.lbvh_build(xyzw, idx)
        for key in ("morton", "order", "child", "parent", "leaf_parent"):
            if not np.array_equal(got[key], want[key]):
                print("scene %d (%d triangles): %s differs" % (i, n, key))
                return 1
        if not np.array_equal(got["box"].view(np.uint32), want["box"].view(np.uint32)):
            print("scene %d (%d triangles): boxes differ" % (i, n))
            return 1
        scene.close()
        total += n
        if (i + 1) % 50 == 0:
            print("%d scenes, %d triangles in all: trees identical so far" % (i + 1, total), flush=True)
    print("%d random scenes (%d triangles in all): every tree identical to the sequential restatement" % (a.scenes, total))
    return 0


if __name__ == "__main__":
    sys.exit(main())
