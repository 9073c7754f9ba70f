"""Where the tile kernel's time goes: A/B the kernel with phases cut off, interleaved in one process.

    python tools/phase_timing.py [--width 1920 --height 1080 --spp 1 --rounds 30] [--frames-in-flight S]

With --frames-in-flight S > 1 the launches of a variant go round-robin to S frame slots (attachments per slot) and the
time per frame is wall time over a batch of launches: what each phase costs in throughput rather than in latency.

The switches live in the high bits of inv_params.flags and make the images wrong; they exist only for this script.
"""
import argparse
import os

os.environ["INV_TIMING_SWITCHES"] = "1"  # the library refuses the switches otherwise
# the switches exist only in the timing build of the library (make -C renderer_amd/csrc timing)
os.environ.setdefault("INVERT_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                     "renderer_amd", "libinvert_hip_timing.so"))
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import objfmt  # noqa: E402
import renderer_amd as R  # noqa: E402

VARIANTS = [("full", 0), ("no exact pass", 1 << 24), ("no phase 3", 1 << 20), ("no phase 2,3", 1 << 21), ("fill only", 1 << 22),
            ("tiles only", 1 << 25), ("lane traversal", R.LANE_TRAVERSAL)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=1)
    ap.add_argument("--rounds", type=int, default=30)
    ap.add_argument("--flags", type=int, default=R.LAMBERT | R.INVERT)
    ap.add_argument("--frames-in-flight", type=int, default=1)
    ap.add_argument("--only", help="run a single variant (for counter passes under rocprofv3)")
    a = ap.parse_args()
    global VARIANTS
    if a.only:
        VARIANTS = [v for v in VARIANTS if v[0] == a.only]
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx = R.Context(0, stream=stream.cuda_stream)
    scene = R.Scene(ctx, R.Mesh.load_obj(objfmt.bunny_obj_path()))
    w, h = a.width, a.height
    buf = torch.empty(3, h * w, dtype=torch.int32, device="cuda")
    cam = R.camera_reference(w, h)
    times = {name: [] for name, _ in VARIANTS}
    S = a.frames_in_flight
    if S > 1:
        import time
        bufs = [torch.empty(3, h * w, dtype=torch.int32, device="cuda") for _ in range(S)]
        ptrs = [[b[i].data_ptr() for i in range(3)] for b in bufs]
        n = 300
        for rnd in range(a.rounds + 3):
            for name, bits in VARIANTS:
                prm = R.make_params(w, h, spp=a.spp, flags=a.flags | bits)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(n):
                    scene.render_in_flight(i % S, cam, prm, *ptrs[i % S])
                torch.cuda.synchronize()
                if rnd >= 3:
                    times[name].append((time.perf_counter() - t0) / n * 1e6)
    for rnd in range(a.rounds + 3 if S == 1 else 0):
        for name, bits in VARIANTS:
            prm = R.make_params(w, h, spp=a.spp, flags=a.flags | bits)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(10):
                scene.render_device(cam, prm, buf[0].data_ptr(), buf[1].data_ptr(), buf[2].data_ptr())
            e1.record(stream)
            e1.synchronize()
            if rnd >= 3:
                times[name].append(e0.elapsed_time(e1) * 100.0)  # us per launch
    for name, _ in VARIANTS:
        t = times[name]
        print("%-16s median %8.2f us   min %8.2f us" % (name, statistics.median(t), min(t)))


if __name__ == "__main__":
    main()
