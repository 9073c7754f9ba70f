"""Host wall time of a scene build (upload, assembly, LBVH) for the bunny and for grids of instances.

    python tools/build_probe.py
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import common  # noqa: E402
import objfmt  # noqa: E402
import renderer_amd as R  # noqa: E402


def main():
    ctx = R.Context(0)
    mesh = R.Mesh.load_obj(objfmt.bunny_obj_path()).pin()
    for side in (1, 2, 4, 8, 16, 32):
        tr = common.grid_translations(side) if side > 1 else None
        times = []
        for _ in range(12):
            t0 = time.perf_counter()
            scene = R.Scene(ctx, mesh, translations=tr)
            times.append((time.perf_counter() - t0) * 1e3)
            ntri = scene.counts()[0]
            scene.close()
        print("%8d triangles: median %.3f ms, min %.3f ms (first %.3f ms)  %.2f G triangles/s"
              % (ntri, float(np.median(times[2:])), min(times[2:]), times[0], ntri / float(np.median(times[2:])) / 1e6))


if __name__ == "__main__":
    main()
