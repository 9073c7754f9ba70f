"""Per-workgroup statistics of the tile kernel: list sizes and in-kernel cycle stamps per phase, for its 8x8 pixel
tiles (what a single draw uses) or its 16x16 pixel blocks (--blocks: what inv_render_batch uses).

    python tools/tile_stats.py [--width 1920 --height 1080] [--blocks] [--instances-side 32]

Uses the diagnostic switches in the high bits of inv_params.flags (images are not produced).
"""
import argparse
import os

os.environ["INV_TIMING_SWITCHES"] = "1"  # the library refuses the switches otherwise
# the switches exist only in the timing build of the library (make -C renderer_amd/csrc timing)
os.environ.setdefault("INVERT_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                     "renderer_amd", "libinvert_hip_timing.so"))
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import common  # noqa: E402
import objfmt  # noqa: E402
import renderer_amd as R  # noqa: E402

SKIP_AFTER2, COUNTS = 1 << 20, 1 << 23


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--blocks", action="store_true")
    ap.add_argument("--instances-side", type=int, default=1, help="a grid of this many by this many copies, seen from afar")
    a = ap.parse_args()
    E = 16 if a.blocks else 8            # edge of a workgroup's pixel square
    mode = R.BLOCKS if a.blocks else 0
    w, h = a.width, a.height
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx = R.Context(0, stream=stream.cuda_stream)
    side = a.instances_side
    scene = R.Scene(ctx, R.Mesh.load_obj(objfmt.bunny_obj_path()),
                    translations=common.grid_translations(side) if side > 1 else None)
    cam = R.camera_reference(w, h)
    if side > 1:
        cam = common.pulled_back_camera(cam, 0.27 * side)
    _, d, p = scene.render(cam, R.make_params(w, h, flags=SKIP_AFTER2 | COUNTS | mode))
    nr, nrec = d.view(np.uint32)[::E, ::E], p[::E, ::E]
    m = (nr < 100000) & (nrec < 100000)
    print(("blocks" if a.blocks else "tiles") + " traced %d; ranges mean %.1f max %d; records mean %.1f max %d" %
          (m.sum(), nr[m].mean(), nr[m].max(), nrec[m].mean(), nrec[m].max()))
    print("records histogram [0,1,8,16,32,64,128,256,...]:", np.histogram(nrec[m], bins=[0, 1, 8, 16, 32, 64, 128, 257, 1 << 30])[0])
    print("ranges histogram [0,1,8,16,32,64,128,256,...]:", np.histogram(nr[m], bins=[0, 1, 8, 16, 32, 64, 128, 257, 1 << 30])[0])
    # stamps of the last of several launches enqueued back to back, as in a stream of frames (a launch on an idle
    # device starts its workgroups over a few microseconds, XCD after XCD)
    buf = torch.empty(3, h * w, dtype=torch.int32, device="cuda")
    # STATS_FLAGS: further switches, e.g. 33554432 = no fill part (do the clear stores slow the traced blocks down?)
    prm = R.make_params(w, h, flags=R.LAMBERT | R.INVERT | COUNTS | mode | int(os.environ.get("STATS_FLAGS", "0")))
    for _ in range(8):
        scene.render_device(cam, prm, buf[0].data_ptr(), buf[1].data_ptr(), buf[2].data_ptr())
    stream.synchronize()
    d = buf[1].cpu().numpy().view(np.uint32).reshape(h, w)
    ys, xs = np.nonzero(m)
    t1, t2, t3, n = (d[::E, k::E][ys, xs].astype(np.int64) for k in range(4))
    for name, t in (("phase 1", t1), ("phase 2", t2), ("phase 3", t3), ("total", t1 + t2 + t3)):
        print("%-8s mean %6.0f  median %6.0f  max %6.0f cycles" % (name, t.mean(), np.median(t), t.max()))
    # phase 1 in parts (second pixel row of the block's first quadrant holds lanes 8..): start-up to the first barrier,
    # cut groups, cut records; and the number of groups that met the block's frustum
    if a.blocks:
        pa, pb, pc, ng = d[::E, 7::E][ys, xs], d[1::E, 0::E][ys, xs], d[1::E, 1::E][ys, xs], d[1::E, 2::E][ys, xs]
        print("phase 1 parts: start %.0f, groups %.0f, records %.0f cycles (mean); groups kept mean %.1f max %d" %
              (pa.mean(), pb.mean(), pc.mean(), ng.mean(), ng.max()))
    real = d[::E, 4::E][ys, xs].astype(np.int64)  # 100 MHz ticks over the three phases
    ok = real > 0
    print("in-kernel clock: %.0f MHz (cycles / constant 100 MHz ticks, median over blocks)" %
          np.median((t1 + t2 + t3)[ok] / real[ok] * 100.0))
    start = d[::E, 6::E][ys, xs].astype(np.int64)
    print("block start spread: %.2f us; last block end - first block start: %.2f us" %
          ((start.max() - start.min()) / 100.0, ((start + real).max() - start.min()) / 100.0))
    rel = (start - start.min()) / 100.0
    print("start (us) percentiles 0..100:   ", " ".join("%.2f" % v for v in np.percentile(rel, range(0, 101, 10))))
    end = rel + real / 100.0
    print("end   (us) percentiles 0..100:   ", " ".join("%.2f" % v for v in np.percentile(end, range(0, 101, 10))))
    i = int(np.argmax(t1 + t2 + t3))
    print("slowest block: %d records, phases %d / %d / %d cycles" % (n[i], t1[i], t2[i], t3[i]))
    for lo, hi in ((0, 1), (1, 8), (8, 16), (16, 32), (32, 256), (256, 1 << 30)):
        k = (n >= lo) & (n < hi)
        if k.any():
            print("records [%3d,%3d): %4d blocks, phases %6.0f / %6.0f / %6.0f cycles" %
                  (lo, hi, k.sum(), t1[k].mean(), t2[k].mean(), t3[k].mean()))


if __name__ == "__main__":
    main()
