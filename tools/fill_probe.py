#!/usr/bin/env python3
"""Time per 1080p frame of batches of four (inv_render_batch) for one build of the library: whole frames and the
clear alone (timing build: PROBE of the fill part), on 2 and 3 frame slots, through `sets` attachment sets per slot
group (1: the sets stay inside the Infinity Cache, 4: they stream to HBM).

    INVERT_HIP_LIB=variants/libinvert_X.so python tools/fill_probe.py label
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["INV_TIMING_SWITCHES"] = "1"
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

import objfmt  # noqa: E402
import renderer_amd as R  # noqa: E402


def main():
    label = sys.argv[1] if len(sys.argv) > 1 else "?"
    ctx = R.Context(0)
    scene = R.Scene(ctx, R.Mesh.load_obj(objfmt.bunny_obj_path()))
    w, h, F = 1920, 1080, 4
    cam = R.camera_reference(w, h)
    out = [label.ljust(10)]
    for M in (1, 4):
        for flags, name in ((0, "full"), (1 << 22, "clear")):
            prm = R.make_params(w, h, flags=R.LAMBERT | R.INVERT | flags)
            for S in (2, 3):
                planes = [torch.empty(3, w * h, dtype=torch.int32, device="cuda") for _ in range(S * F * M)]
                batches = [scene.make_batch([cam] * F, [prm] * F, [(p[0].data_ptr(), p[1].data_ptr(), p[2].data_ptr())
                                                                     for p in planes[j * F:(j + 1) * F]]) for j in range(S * M)]
                K, best = 600, 1e9
                for rep in range(4):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for i in range(K):
                        scene.render_batch(batches[i % (S * M)], i % S)
                    torch.cuda.synchronize()
                    best = min(best, (time.perf_counter() - t0) / (K * F) * 1e6)
                out.append("sets x%d %s x%d %5.2f" % (M, name, S, best))
                del planes
    print("  ".join(out), flush=True)
    scene.close()
    ctx.close()


if __name__ == "__main__":
    main()
