#!/usr/bin/env python3
"""How much does keeping several frames in flight buy, and what does a draw cost on the host?

The reference keeps two frames in flight (MAX_FRAMES_IN_FLIGHT, src/invert.cpp:241).  This probe renders the 1080p
bunny K times over S frame slots (inv_render_in_flight, attachments per slot) for S = 1..4 and prints the time per
frame; then the same loop on a 32x32 frame, where the GPU work is negligible, which leaves the host's cost per
inv_render_in_flight call.
"""
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
    K = 4000
    ctx = R.Context(0)
    scene = R.Scene(ctx, R.Mesh.load_obj(objfmt.bunny_obj_path()))
    for (w, h) in ((1920, 1080), (32, 32)):
        cam = R.camera_reference(w, h)
        prm = R.make_params(w, h, flags=R.LAMBERT | R.INVERT)
        for S in (1, 2, 3, 4):
            planes = [torch.empty(3, w * h, dtype=torch.int32, device="cuda") for _ in range(S)]
            ptrs = [[p[i].data_ptr() for i in range(3)] for p in planes]
            for rep in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(K):
                    scene.render_in_flight(i % S, cam, prm, *ptrs[i % S])
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            same = all(torch.equal(planes[0], p) for p in planes[1:])
            print("%dx%d, %d in flight: %.2f us per frame (%.0f Grays/s); host loop %.2f us per call; slots identical: %s"
                  % (w, h, S, dt / K * 1e6, w * h * K / dt / 1e9, (t1 - t0) / K * 1e6, same), flush=True)
    scene.close()
    ctx.close()


if __name__ == "__main__":
    main()
