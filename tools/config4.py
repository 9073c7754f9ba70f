"""Config 4: 1024 bunny instances (32 x 32 grid, 5 087 232 triangles) flattened into one LBVH, 1080p.

    python tools/config4.py [--side 32] [--check]

Times the scene build (upload, assembly, LBVH) and the render separately; --check compares the frame with the CPU
oracle (takes a while: the oracle builds its own LBVH over 5 M triangles).
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import common  # noqa: E402
import objfmt  # noqa: E402
import renderer_amd as R  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", type=int, default=32)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--builds", type=int, default=8)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    a = ap.parse_args()
    w, h = a.width, a.height
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx = R.Context(0, stream=stream.cuda_stream)
    mesh = R.Mesh.load_obj(objfmt.bunny_obj_path()).pin()
    tr = common.grid_translations(a.side)
    times = []
    for rep in range(a.builds):
        t0 = time.perf_counter()
        scene = R.Scene(ctx, mesh, translations=tr)
        times.append((time.perf_counter() - t0) * 1e3)
        ntri, nnodes, nbytes = scene.counts()
        if rep < a.builds - 1:
            scene.close()
    print("build: %d triangles, %.1f MB scene; host wall including upload and the final sync: first %.2f ms, then "
          "median %.2f ms, min %.2f ms over %d builds" % (ntri, nbytes / 1e6, times[0], float(np.median(times[1:])),
                                                          min(times[1:]), len(times) - 1))
    cam = common.grid_camera(R.camera_reference(w, h), a.side)
    buf = torch.empty(3, h * w, dtype=torch.int32, device="cuda")
    rows = [("auto (one sample)", 0, 1), ("through the LBVH (INV_TILE_TRAVERSAL)", R.TILE_TRAVERSAL, 1),
            ("auto (two samples: the tile kernel)", 0, 2)]
    if a.side <= 40:
        rows.append(("lane traversal", R.LANE_TRAVERSAL, 1))
    for name, flags, spp in rows:
        prm = R.make_params(w, h, spp=spp, seed=1, flags=R.LAMBERT | R.INVERT | flags)
        path = scene.draw_path(cam, prm)
        name = "%s [path %d%s]" % (name, path & 15, ", binned" if path & R.api.PATH_BINNED else "")
        for _ in range(2):
            scene.render_device(cam, prm, buf[0].data_ptr(), buf[1].data_ptr(), buf[2].data_ptr())
        stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(5):
            scene.render_device(cam, prm, buf[0].data_ptr(), buf[1].data_ptr(), buf[2].data_ptr())
        e1.record(stream)
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 5
        hits = int((buf[2] != -1).sum())
        print("%-48s %.3f ms per frame, %.1f Mrays/s, %d hit pixels" % (name, ms, w * h / ms / 1e3, hits))
    if a.check:
        from oracle import oracle as O
        xyzw, idx = O.load_obj(objfmt.bunny_obj_path())
        fx, fi = common.flatten_instances(xyzw, idx, tr)
        prm = R.make_params(w, h, flags=R.LAMBERT | R.INVERT)
        got = scene.render(cam, prm)
        t0 = time.perf_counter()
        want = O.render_bvh(fx, fi, O.Camera.from_buffer_copy(bytes(cam)), O.make_params(w, h, flags=R.LAMBERT | R.INVERT))
        print("oracle %.1f s" % (time.perf_counter() - t0))
        common.assert_frames_equal(got, want, "config 4")
        print("config 4 frame is bit-exact against the oracle")


if __name__ == "__main__":
    main()
