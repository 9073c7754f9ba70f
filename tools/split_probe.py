#!/usr/bin/env python3
"""Experiment: the clear (fill) and the traced blocks of the headline frames as separate launches on separate streams.

Uses the timing build's switches to cut a batch launch down to one of its two parts: slots 0/1 submit "fill only"
launches, slots 2/3 "trace only" launches, with no ordering between them, and the time per frame is compared with
the usual launches (both parts in one grid) on two slots.  SETS attachment sets per slot pair (1: Infinity-Cache
resident, 4: streaming).
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["INV_TIMING_SWITCHES"] = "1"
os.environ.setdefault("INVERT_HIP_LIB", os.path.join(ROOT, "renderer_amd", "libinvert_hip_timing.so"))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

import objfmt  # noqa: E402
import renderer_amd as R  # noqa: E402

SKIP_ALL, SKIP_FILL = 1 << 22, 1 << 25


def main():
    ctx = R.Context(0)
    scene = R.Scene(ctx, R.Mesh.load_obj(objfmt.bunny_obj_path()))
    w, h = 1920, 1080
    cam = R.camera_reference(w, h)
    base = R.LAMBERT | R.INVERT
    for F in (4, 8):
        for M in (1, 4):
            nsets = 2 * F * M
            planes = [torch.empty(3, w * h, dtype=torch.int32, device="cuda") for _ in range(nsets)]
            def batches(flags):
                prm = R.make_params(w, h, flags=base | flags)
                return [scene.make_batch([cam] * F, [prm] * F, [(p[0].data_ptr(), p[1].data_ptr(), p[2].data_ptr())
                                                                 for p in planes[j * F:(j + 1) * F]]) for j in range(2 * M)]
            full, fill, trace = batches(0), batches(SKIP_ALL), batches(SKIP_FILL)
            K = 2400 // F
            res = {}
            for name in ("one launch, 2 slots", "fill slot 0 + trace slot 1", "fill 2 slots + trace 1 slot", "fill only", "trace only 2 slots", "trace only 1 slot"):
                best = 1e9
                for rep in range(4):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for i in range(K):
                        j = i % (2 * M)
                        if name.startswith("one"):
                            scene.render_batch(full[j], i % 2)
                        elif name == "fill only":
                            scene.render_batch(fill[j], i % 2)
                        elif name.startswith("trace only"):
                            scene.render_batch(trace[j], i % 2 if "2 slots" in name else 0)
                        elif name.startswith("fill slot 0"):
                            scene.render_batch(fill[j], 0)
                            scene.render_batch(trace[j], 1)
                        else:
                            scene.render_batch(fill[j], i % 2)
                            scene.render_batch(trace[j], 2)
                    torch.cuda.synchronize()
                    best = min(best, (time.perf_counter() - t0) / (K * F) * 1e6)
                res[name] = best
            print("F=%d sets=%d (%.0f MB): " % (F, nsets, nsets * 24.9) + "  ".join("%s %.2f us" % kv for kv in res.items()), flush=True)
    scene.close()
    ctx.close()


if __name__ == "__main__":
    main()
