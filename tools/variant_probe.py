#!/usr/bin/env python3
"""Time per frame of the bench workloads for one build of the library (INVERT_HIP_LIB names it), with 1 and 2
frames in flight, and check the frames against the default build's.  Tuning experiments only.

    INVERT_HIP_LIB=variants/libinvert_X.so python tools/variant_probe.py [label]
"""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

import objfmt  # noqa: E402
import renderer_amd as R  # noqa: E402


def main():
    label = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(R.api.LIB_PATH)
    ctx = R.Context(0)
    scene = R.Scene(ctx, R.Mesh.load_obj(objfmt.bunny_obj_path()))
    line = [label.ljust(12)]
    for name, w, h, spp, ao, K in (("1080p", 1920, 1080, 1, 0, 3000), ("4k16", 3840, 2160, 16, 0, 600), ("ao64", 1920, 1080, 1, 64, 300)):
        cam = R.camera_reference(w, h)
        # PROBE_FLAGS: extra flag bits, e.g. 4194304 (with INV_TIMING_SWITCHES=1) skips the traced tiles: clear only
        prm = R.make_params(w, h, spp=spp, ao_rays=ao, flags=R.LAMBERT | R.INVERT | int(os.environ.get("PROBE_FLAGS", "0")))
        for S in (1, 2, 3, 4):
            planes = [torch.empty(3, w * h, dtype=torch.int32, device="cuda") for _ in range(S)]
            ptrs = [[p[i].data_ptr() for i in range(3)] for p in planes]
            best = 1e9
            for rep in range(4):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(K):
                    scene.render_in_flight(i % S, cam, prm, *ptrs[i % S])
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / K * 1e6)
            line.append("%s x%d %7.2f us" % (name, S, best))
        line.append(hashlib.sha1(planes[0].cpu().numpy().tobytes()).hexdigest()[:8])
    print("  ".join(line), flush=True)
    # several frames per launch (inv_render_batch), each into attachments of its own
    w, h = 1920, 1080
    cam, prm = R.camera_reference(w, h), R.make_params(w, h, flags=R.LAMBERT | R.INVERT | int(os.environ.get("PROBE_FLAGS", "0")))
    for F in [int(v) for v in os.environ.get("PROBE_FRAMES", "2,4,8").split(",")]:
        line = ["  %d frames per launch:" % F]
        # PROBE_SETS = M: every slot rotates through M groups of F attachment sets (the footprint of the frames in
        # flight against the 256 MB Infinity Cache: one 1080p set is 24.9 MB)
        M = int(os.environ.get("PROBE_SETS", "1"))
        for S in (1, 2, 3):
            planes = [torch.empty(3, w * h, dtype=torch.int32, device="cuda") for _ in range(S * F * M)]
            batches = [scene.make_batch([cam] * F, [prm] * F, [(p[0].data_ptr(), p[1].data_ptr(), p[2].data_ptr())
                                                                 for p in planes[j * F:(j + 1) * F]]) for j in range(S * M)]
            K, best = 2400 // F, 1e9
            for rep in range(4):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(K):
                    scene.render_batch(batches[i % (S * M)], i % S)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / (K * F) * 1e6)
            line.append("x%d %6.2f us" % (S, best))
        line.append(hashlib.sha1(planes[-1].cpu().numpy().tobytes()).hexdigest()[:8])
        print("  ".join(line), flush=True)
    scene.close()
    ctx.close()


if __name__ == "__main__":
    main()
