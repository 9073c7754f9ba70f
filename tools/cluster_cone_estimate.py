"""How many groups of 16 Morton-consecutive triangles of the instanced scenes face away from the eye as a whole.

    python tools/cluster_cone_estimate.py [--sides 32 118]        (CPU only: numpy and the oracle's sequential LBVH build)

For every group of 16 consecutive triangles of the sorted order: the cone of its normals (axis = normalised sum, half
angle = widest normal), the cone test a cluster-culling pass would use (with that pass's margins), and the exact
answer (all 16 planes behind the eye).  Printed: the share of groups each drops, the share of back-facing triangles
and the median half angle.  This is the measurement that decided against dropping whole groups in the
one-lane-per-triangle pass.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import common  # noqa: E402
import objfmt  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sides", type=int, nargs="+", default=[32, 118])
    a = ap.parse_args()
    xyzw, idx = O.load_obj(objfmt.bunny_obj_path())
    for side in a.sides:
        fx, fi = common.flatten_instances(xyzw, idx, common.grid_translations(side))
        order = O.lbvh_build(fx, fi)["order"].astype(np.int64)
        n16 = (len(order) // 16) * 16
        tri = fx[:, :3][fi.reshape(-1, 3)[order[:n16]].astype(np.int64)].astype(np.float64)
        nrm = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        eye = np.array([0.0, 0.11, 0.27 * side])  # the grid camera's eye in object space (tests/common.py: grid_camera)
        c = tri.reshape(-1, 16, 3, 3)
        lo, hi = c.min(axis=(1, 2)), c.max(axis=(1, 2))
        cen, rad = (lo + hi) / 2, np.linalg.norm((hi - lo) / 2, axis=1)
        nn = nrm.reshape(-1, 16, 3)
        ax = nn.sum(axis=1)
        ax /= np.linalg.norm(ax, axis=1, keepdims=True)
        cmin = (nn * ax[:, None, :]).sum(axis=2).min(axis=1)
        sine = np.sqrt(np.maximum(0, 1 - cmin ** 2)) + 2e-3
        v = cen - eye
        dist = np.linalg.norm(v, axis=1)
        away = ((v * ax).sum(axis=1) >= (sine + 4e-3) * dist + rad) & (cmin > 0.1)
        back = (((tri[:, 0] - eye) * nrm).sum(axis=1) > 0).reshape(-1, 16)
        print("%d x %d instances, %d groups of 16: the cone test drops %.1f %%, an exact all-sixteen test %.1f %%; "
              "%.1f %% of the triangles face away; median cone half angle %.0f degrees"
              % (side, side, len(c), 100 * away.mean(), 100 * back.all(axis=1).mean(), 100 * back.mean(),
                 np.degrees(np.arcsin(np.clip(np.median(sine), 0, 1)))))


if __name__ == "__main__":
    main()
