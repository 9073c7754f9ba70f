#!/usr/bin/env python3
"""Scene build of a large flat mesh (1024 bunnies flattened on the host: 2.56 M vertices, 5.09 M triangles, 102 MB of
vertex and index data to upload) against the instanced build of the same triangles (one bunny and 1024 translations to
upload): what the host-to-HBM copy costs next to the LBVH build."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402,F401

import common  # noqa: E402
import objfmt  # noqa: E402
import renderer_amd as R  # noqa: E402


def main():
    mesh = R.Mesh.load_obj(objfmt.bunny_obj_path())
    tr = common.grid_translations(32, 0.2)
    xyzw, idx = common.flatten_instances(mesh.vertices, mesh.indices, tr)
    flat = R.Mesh.from_arrays(xyzw, idx)
    ctx = R.Context(0)
    t0 = time.perf_counter()
    pinned = R.Mesh.from_arrays(xyzw, idx).pin()
    print("inv_mesh_from_arrays + inv_mesh_pin: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
    for name, build in (("instanced", lambda: R.Scene(ctx, mesh, tr)), ("flat", lambda: R.Scene(ctx, flat)),
                        ("flat pinned", lambda: R.Scene(ctx, pinned))):
        for rep in range(4):
            ctx.synchronize()
            t0 = time.perf_counter()
            scene = build()
            dt = time.perf_counter() - t0
            ntri = scene.counts()[0]
            scene.close()
            print("%-12s build %d: %d triangles, %.2f ms host wall" % (name, rep, ntri, dt * 1e3), flush=True)
    print("upload: %.1f MB vertices + %.1f MB indices" % (xyzw.nbytes / 1e6, idx.nbytes / 1e6))
    ctx.close()


if __name__ == "__main__":
    main()
