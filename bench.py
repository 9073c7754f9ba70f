#!/usr/bin/env python3
"""Headline benchmark: Mrays/s on the 1080p bunny (primary rays + Lambert shade + invert post-pass).

Every frame of the timed region is written to HBM: the frames in flight rotate through enough attachment sets
(over 1 GB, four times the 256 MB Infinity Cache in front of HBM) that no set is still cached when it is drawn into
again, so `value` and `roofline` are HBM figures.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

One step = one batch of --frames-per-step (default 256 for the headline, 64 for the heavier workloads) frames of the workload rendered through the C ABI
(include/invert.h) with every input (triangles, LBVH) and output buffer resident in HBM; a frame of the headline
workload lasts a few microseconds, so a single frame per step would leave a handful of steps too short to time.  On one GPU the frames are submitted --frames-per-launch at a time
(inv_render_batch, default 4: one launch of the tile kernel draws them all) and the submissions go round-robin to
--frames-in-flight frame slots (default 2, the reference's MAX_FRAMES_IN_FLIGHT, src/invert.cpp:241), whose launches
overlap on the GPU.  Every frame is a whole draw into attachments of its own (slots x frames-per-launch sets).
Workloads with occlusion rays or four and more samples per pixel are drawn one frame per launch.  With N > 1 (launched by torch.distributed.run, one
process per GPU) 32x32 screen tiles are dealt round-robin to the ranks, every rank renders its tiles, and the
three attachments are gathered to rank 0 over RCCL and put together there; the same frame is rendered whatever N
is, so scaling is "strong".  The sparse composite works in groups of --composite-frames frames (default 8): one
submission of draws and one gather per group, two sets of buffers, so that a group's gather overlaps the next
group's draws (1: one frame per step on the context's stream).  --composite sparse (default) deals out only the tiles under the scene's projected
bounds and rank 0 clears the rest of its frame itself; --composite dense deals out and gathers the whole frame.

Rank 0 prints one JSON line with the contract fields plus
  roofline      -- the render kernel against the HBM roofline (algorithmic bytes = 12 B per pixel written +
                   the scene read once), duration from HIP events around every launch of the timed region
  cpu_baseline  -- the CPU oracle (a scalar C++ port of the same algorithm; the reference has no CPU path)
                   timed on this host's cores on the same frame
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def _launch_ranks():
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: start the N ranks from here.

    This runs before torch, the library or anything else that could open the GPU is imported: the parent only starts
    other processes and never touches a GPU itself.  By default it hands the job to torch's own launcher
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>`),
    which keeps the rendezvous store and ends every worker when one fails; INV_BENCH_LAUNCHER=spawn starts N
    interpreters of this file directly instead (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, each
    in a session of its own; a rank that fails takes the others down after five seconds).  Either way only the JSON
    line of rank 0 reaches standard output (what libraries print there goes to standard error) and the exit status
    is the worst of the ranks'.  Returns None when there is nothing to launch."""
    if "WORLD_SIZE" in os.environ:
        return None
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--gpus", type=int, default=1)
    known, _ = ap.parse_known_args()
    if known.gpus <= 1 or any(a in ("-h", "--help") for a in sys.argv[1:]):
        return None
    import signal
    import socket
    import subprocess
    import threading
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
    variant = os.environ.get("INV_BENCH_LAUNCHER", "torchrun")

    def forward(stream):  # the JSON line to standard output, anything else a library printed there to standard error
        for raw in stream:
            line = raw.decode("utf-8", "replace")
            (sys.stdout if line.startswith("{") else sys.stderr).write(line)
        sys.stdout.flush()

    if variant == "torchrun":
        # torch's own launcher does the rest: one worker per rank, the rendezvous store in the launcher, and the end of
        # every worker when one of them fails
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(known.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE)
        pump = threading.Thread(target=forward, args=(p.stdout,), daemon=True)
        pump.start()
        try:
            rc = p.wait()
        except KeyboardInterrupt:
            p.send_signal(signal.SIGINT)
            rc = p.wait()
        pump.join(timeout=10)
        return rc if rc >= 0 else 128 - rc
    procs = []
    for r in range(known.gpus):
        renv = dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(known.gpus), MASTER_ADDR="127.0.0.1",
                    MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + sys.argv[1:], env=renv,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr,
                                      start_new_session=True))
    pump = threading.Thread(target=forward, args=(procs[0].stdout,), daemon=True)
    pump.start()
    worst, failed_at = 0, None
    try:
        while any(p.poll() is None for p in procs):
            time.sleep(0.05)
            for r, p in enumerate(procs):
                rc = p.poll()
                if rc not in (None, 0) and failed_at is None:
                    failed_at = time.monotonic()
                    print("bench.py: rank %d exited with status %d; stopping the other ranks" % (r, rc), file=sys.stderr)
            if failed_at is not None and time.monotonic() - failed_at > 5.0:  # the others had their chance to follow
                for p in procs:
                    if p.poll() is None:
                        p.send_signal(signal.SIGTERM)
                failed_at = float("inf")
    except KeyboardInterrupt:
        for p in procs:
            if p.poll() is None:
                p.send_signal(signal.SIGTERM)
    for p in procs:
        try:
            rc = p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
            rc = p.wait()
        if rc != 0:
            worst = max(worst, rc if rc > 0 else 128 - rc)
    pump.join(timeout=10)
    return worst


if __name__ == "__main__":
    _status = _launch_ranks()
    if _status is not None:
        sys.exit(_status)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import common  # noqa: E402
import objfmt  # noqa: E402
import renderer_amd as R  # noqa: E402
from renderer_amd import dist as RD  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
VALU_PEAK_TFLOPS = 157.3  # fp32 vector: 256 CUs x 4 SIMDs x 16 lanes x 2 flops (FMA) x 2.4 GHz
# algorithmic flops of the per-ray steps (counted on the CPU oracle's own traversal, not on the GPU)
FLOPS_RAY, FLOPS_BOX, FLOPS_TRI, FLOPS_SHADE = 20, 22, 51, 15

WORKLOADS = {
    # the configuration the headline metric is quoted on
    "bunny_1080p_primary_lambert_invert": dict(width=1920, height=1080, spp=1, flags=R.LAMBERT | R.INVERT),
    # 4K, 16 jittered samples per pixel
    "bunny_4k_16spp": dict(width=3840, height=2160, spp=16, flags=R.LAMBERT | R.INVERT),
    # 64 occlusion rays per hit pixel on top of the headline configuration
    "bunny_1080p_ao64": dict(width=1920, height=1080, spp=1, ao_rays=64, flags=R.LAMBERT | R.INVERT),
    # the same from close by (eye 0.3 from the model instead of 1): the bunny fills the frame, so the occlusion rays
    # (64 per hit pixel) are most of the work
    "bunny_1080p_ao64_closeup": dict(width=1920, height=1080, spp=1, ao_rays=64, flags=R.LAMBERT | R.INVERT,
                                     view_translation=(0.0, 0.11, -0.3)),
    # the model turning at 90 degrees per second, one frame per 1/60 s: a stream of different frames from one scene
    # build (one GPU only)
    "bunny_1080p_turntable": dict(width=1920, height=1080, spp=1, flags=R.LAMBERT | R.INVERT, turntable=True),
    # 32 x 32 translated copies of the bunny flattened into one LBVH (5 087 232 triangles), seen from far enough to
    # show them all: stresses the build and the scene's HBM traffic; config.build_ms is the scene build
    "bunny_x1024_1080p": dict(width=1920, height=1080, spp=1, flags=R.LAMBERT | R.INVERT, instances_side=32),
    # 118 x 118 = 13 924 copies: 69 174 432 triangles, the size BASELINE.json's fourth configuration states ("~70M
    # tris"; the mesh in res/ has 4 968 triangles, so 1 024 copies of it are 5.09 M); far plane at 100 instead of 10
    "bunny_x13924_1080p": dict(width=1920, height=1080, spp=1, flags=R.LAMBERT | R.INVERT, instances_side=118),
    # the headline extent in parity mode (what the reference's shaders draw: no Lambert, no invert)
    "bunny_1080p_parity": dict(width=1920, height=1080, spp=1, flags=0),
}


def pmc_traffic(workload, world, kernel, frames_per_launch):
    """HBM bytes per launch of the render kernel from the newest committed rocprofv3 PMC summary, or None.

    The counters need separate profiler passes (FETCH_SIZE and WRITE_SIZE do not fit one), so they cannot be read
    inside this process; tools/summarize_profiles.py writes profiles/*_pmc.json from those passes."""
    import glob
    if world != 1:
        return None
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json"))):
        try:
            doc = json.load(open(path))
        except (OSError, ValueError):
            continue
        k = doc.get("kernels", {}).get(kernel)
        if doc.get("workload") == workload and k and doc.get("frames_per_launch", 1) == frames_per_launch:
            best = k["hbm_bytes_per_launch"]
    return best


def pmc_valu(workload, world, kernel):
    """VALU utilisation of `kernel` from the newest profiles/*_valu.json of this workload (tools/valu_summary.py
    writes it from a rocprofv3 --pmc pass of this bench), or None."""
    import glob
    if world != 1:
        return None
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_valu.json"))):
        try:
            doc = json.load(open(path))
        except (OSError, ValueError):
            continue
        k = doc.get("workloads", {}).get(workload, {}).get(kernel)
        if k:
            best = dict(k, source=os.path.basename(path))
    return best


def cpu_baseline(wl, camera, budget_s=10.0):
    """Time the CPU oracle on the same frame (bounded: whole frames until the budget is used)."""
    from oracle import oracle as O

    xyzw, idx = O.load_obj(objfmt.bunny_obj_path())
    if "instances_side" in wl:
        xyzw, idx = common.flatten_instances(xyzw, idx, common.grid_translations(wl["instances_side"]))
    w, h = wl["width"], wl["height"]
    cam = O.Camera.from_buffer_copy(bytes(camera))  # the same 204 bytes the GPU drew with
    prm = O.make_params(w, h, spp=wl["spp"], flags=wl["flags"], ao_rays=wl.get("ao_rays", 0))
    # threads used: the box gives one GPU's job a share of 16 CPUs, whatever the machine has (cores_present)
    cores_present = os.cpu_count() or 1
    cores = min(16, cores_present)
    cnt = O.Counters()
    t0 = time.perf_counter()
    out = O.render_bvh(xyzw, idx, cam, prm, nthreads=cores, counters=cnt)  # warm; also counts the traversal's tests
    frames, dt = 1, time.perf_counter() - t0
    if dt <= budget_s:  # (a scene whose one frame takes longer than the whole budget: that frame is the sample)
        frames, t0 = 0, time.perf_counter()
        while True:
            out = O.render_bvh(xyzw, idx, cam, prm, nthreads=cores)
            frames += 1
            dt = time.perf_counter() - t0
            if dt > budget_s or frames >= 1024:  # about ten seconds of wall time on the host's cores
                break
    hits = int((out[2] != 0xFFFFFFFF).sum())
    rays = (w * h * wl["spp"] + wl.get("ao_rays", 0) * hits) * frames
    flops = (FLOPS_RAY * cnt.rays + FLOPS_BOX * cnt.box_tests + FLOPS_TRI * cnt.tri_tests + FLOPS_SHADE * cnt.hits)
    return dict(value=rays / dt / 1e6, unit="Mrays/s", cores=cores, cores_present=cores_present, kind="port",
                sample="%d whole %dx%d frames (%d spp) through the oracle's CPU LBVH, %d threads, %.1f s"
                       % (frames, w, h, wl["spp"], cores, dt)), out, dict(
        rays=int(cnt.rays), box_tests=int(cnt.box_tests), tri_tests=int(cnt.tri_tests), hits=int(cnt.hits), flops=int(flops))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=60)
    ap.add_argument("--frames-per-step", type=int, default=None,
                    help="a step is one batch of this many frames of the workload (a single frame lasts a few "
                         "microseconds: too little to time a handful of steps); default 256 for the workloads whose "
                         "frames share launches, 64 for the heavier ones")
    ap.add_argument("--workload", default="bunny_1080p_primary_lambert_invert", choices=sorted(WORKLOADS))
    ap.add_argument("--frames-in-flight", type=int, default=None, choices=range(1, R.api.MAX_FRAMES_IN_FLIGHT + 1),
                    help="one GPU: frame slots the submissions are dealt to (default: 3 where frames share launches, "
                         "2 where every frame is a launch of its own)")
    ap.add_argument("--frames-per-launch", type=int, default=4, choices=range(1, 9),
                    help="one GPU: frames per inv_render_batch submission (1 with --frames-in-flight 1: plain inv_render "
                         "on the context's stream)")
    ap.add_argument("--composite-frames", type=int, default=8, choices=range(1, 9),
                    help="N > 1, sparse composite: frames per submission and per gather")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=10.0,
                    help="wall time the CPU oracle is given for its whole frames (the bounded sample of the baseline)")
    ap.add_argument("--no-extra-passes", action="store_true",
                    help="skip the passes after the timed region (roofline.alone, roofline.infinity_cache_resident): "
                         "under a profiler the kernel statistics then cover the timed region and its warm-up only")
    ap.add_argument("--attachment-gb", type=float, default=1.0,
                    help="one GPU: the frames in flight rotate through at least this much attachment memory (the "
                         "Infinity Cache holds 0.256 GB), so that every frame is written to HBM")
    ap.add_argument("--warmup-ms", type=float, default=50.0,
                    help="warm up for at least this long (in addition to --warmup steps): a handful of steps of a few "
                         "hundred microseconds would time the clock ramp")
    ap.add_argument("--brute-force", action="store_true", help="time the brute-force kernel instead of the LBVH one")
    ap.add_argument("--composite", default="sparse", choices=["sparse", "dense"],
                    help="N > 1: gather only the tiles that can contain hits (sparse) or the whole frame (dense)")
    ap.add_argument("--root-share", default="auto", choices=["auto", "equal", "none"],
                    help="N >= 3: rank 0 draws an equal share of the tiles, or none (INV_ROOT_IDLE: it alone writes the "
                         "background and receives the tiles); auto: none from 4 ranks on for frames without occlusion rays")
    ap.add_argument("--gather", default="torch", choices=["torch", "c"],
                    help="N > 1, sparse composite: torch.distributed.gather, or the library's own exchange (inv_comm_gather: "
                         "grouped RCCL send / receive on the context's stream; staging files with --backend gloo)")
    ap.add_argument("--no-extra-workloads", action="store_true",
                    help="N > 1: skip the short passes over the heavier workloads (4K with 16 samples, occlusion rays "
                         "from close by) that follow the timed region and show where sharding pays")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the N > 1 path on one GPU (ranks share cuda:0, gather through host memory)")
    ap.add_argument("--extras-watchdog-seconds", type=float, default=240.0,
                    help="N > 1: the passes over the heavier workloads that follow the timed region are given up after this "
                         "long; the line is then printed without them, with a note (0: off)")
    ap.add_argument("--watchdog-seconds", type=int, default=900,
                    help="a rank still running after this long writes every thread's Python stack to standard error and "
                         "exits with a failure, so that a collective that never completes ends the run with a message "
                         "instead of hanging it (0: off)")
    args = ap.parse_args()
    watched_streams = {}  # name -> torch stream, for the watchdog's report
    if args.watchdog_seconds > 0:
        import faulthandler
        import threading

        def give_up():
            sys.stderr.write("bench.py: rank %s still running after %d s; stacks of its threads:\n"
                             % (os.environ.get("RANK", "0"), args.watchdog_seconds))
            faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
            for name, q in list(watched_streams.items()):  # which of the streams still has work that has not finished
                try:
                    sys.stderr.write("bench.py: rank %s stream %s: %s\n" % (os.environ.get("RANK", "0"), name,
                                                                           "idle" if q.query() else "BUSY"))
                except Exception as exc:  # noqa: BLE001
                    sys.stderr.write("bench.py: stream %s: query failed: %r\n" % (name, exc))
            sys.stderr.flush()
            os._exit(3)

        watchdog = threading.Timer(args.watchdog_seconds, give_up)
        watchdog.daemon = True
        watchdog.start()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:  # (without WORLD_SIZE and with --gpus N > 1 the ranks were started by _launch_ranks above)
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path to time")
    if args.backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit("--gpus %d over RCCL needs %d GPUs, this node shows %d (--backend gloo rehearses the ranks on one GPU)"
                         % (world, world, torch.cuda.device_count()))
    local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    wl = WORKLOADS[args.workload]
    if args.frames_per_step is None:
        # Frames of a few microseconds: 256 per step, so that 20 steps last some 20 ms and the ramp at the start of the
        # timed region and the drain at its end (the slots' last launches do not overlap) weigh under 1 %; with 64 a
        # run of 20 steps read 3.76 us per frame where a long one reads 3.67.  Frames of tens of microseconds and more: 64.
        light = wl["spp"] < 4 and not wl.get("ao_rays", 0) and wl.get("instances_side", 1) == 1
        args.frames_per_step = 256 if light else 64
    nframes, nwarm = args.steps * args.frames_per_step, args.warmup * args.frames_per_step  # frames timed / warmed up
    w, h, spp = wl["width"], wl["height"], wl["spp"]
    flags = wl["flags"] | (R.BRUTE_FORCE if args.brute_force else 0)
    tile = 64 if world == 1 else 32  # several ranks: 32-pixel tiles spread the hit pixels more evenly (1.16 vs 1.62 max/mean at 8)
    # A stream of our own, made torch's current stream: the library launches on it, RCCL orders against it and
    # torch.cuda.Event records on it (a NULL handle would make the library create a private stream).
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    ctx = R.Context(local_rank, stream=stream.cuda_stream)
    watched_streams["context"] = stream
    for j in range(2):
        watched_streams["frame slot %d" % j] = torch.cuda.ExternalStream(ctx.frame_stream(j))
    mesh = R.Mesh.load_obj(objfmt.bunny_obj_path()).pin()  # page-locked arrays: the build's uploads are DMA out of them
    side = wl.get("instances_side", 1)
    translations = common.grid_translations(side) if side > 1 else None
    build_ms = []
    for _ in range(6):  # every rank holds the whole scene; the first build also pays for the allocations
        t_build = time.perf_counter()
        scene = R.Scene(ctx, mesh, translations=translations)
        build_ms.append((time.perf_counter() - t_build) * 1e3)
        if len(build_ms) < 6:
            scene.close()
    ntri, nnodes, scene_bytes = scene.counts()
    cam = R.camera_reference(w, h)
    if side > 1:  # pulled back and raised until the whole grid is in view
        cam = common.grid_camera(cam, side)
    if "view_translation" in wl:  # the reference camera moved: same roll and lens, another eye position
        cam.view[12], cam.view[13], cam.view[14] = wl["view_translation"]
    turntable = bool(wl.get("turntable"))
    if turntable and world > 1:
        raise SystemExit("the turntable workload runs on one GPU")
    cams = [R.camera_turntable(w, h, i / 60.0) for i in range(240)] if turntable else None  # one full turn
    ao_rays = wl.get("ao_rays", 0)
    sparse = world > 1 and args.composite == "sparse"
    idle = world >= 3 and (args.root_share == "none" or (args.root_share == "auto" and world >= 4 and not ao_rays))
    sflags = flags | (R.ROOT_IDLE if idle else 0)  # flags of the sharded draws
    gathered = frame = sc = None
    # frames per submission: occlusion rays and many samples per pixel are drawn frame by frame anyway
    fpl = args.frames_per_launch if world == 1 and not ao_rays and spp < 4 and side == 1 else 1
    fif = (args.frames_in_flight or (3 if fpl > 1 else 2)) if world == 1 else 1  # the composite of several ranks runs on the context's stream
    slots_used = fif > 1 or fpl > 1  # else: plain inv_render on the context's stream
    cb = None
    grouped = sparse and args.composite_frames > 1
    comm = None
    if sparse and args.gather == "c":
        comm = RD.make_comm(ctx, rank, world, "rccl" if args.backend == "nccl" else "host")
    direct = args.backend == "nccl" or comm is not None  # the composite's own exchange moves the tiles
    if grouped:
        fpl = args.composite_frames
        cb = RD.SparseCompositeBatch(scene, cam, w, h, rank, world, frames=fpl, sets=2, tile=tile, spp=spp, flags=sflags,
                                     ao_rays=ao_rays, comm=comm)
        npix = cb.slot_elems if rank else w * h - (world - 1) * cb.slot_elems  # pixels per frame this rank writes
    elif sparse:
        sc = RD.SparseComposite(scene, cam, w, h, rank, world, tile=tile, spp=spp, flags=sflags, ao_rays=ao_rays, comm=comm)
        frame, gathered = sc.frame, sc.gathered
        # pixels this rank's render launch writes
        npix = sc.slot_elems if rank else w * h - (world - 1) * sc.slot_elems
    else:
        prm = R.make_params(w, h, spp=spp, flags=sflags if world > 1 else flags, tile=tile, rank=rank, nranks=world, ao_rays=ao_rays)
        npix, _ = R.tile_layout(prm)
        if world > 1 and npix == 0:  # rank 0 without tiles still brings a share of the gather's size
            npix, _ = R.tile_layout(R.make_params(w, h, spp=spp, flags=sflags, tile=tile, rank=1, nranks=world, ao_rays=ao_rays))
        # attachments: one set (three planes: colour, depth, primitive id; 4 bytes per pixel each) per frame in
        # flight, times `rotate`: every slot goes round `rotate` groups of sets, so that the job streams through
        # --attachment-gb of memory and no set is still in the Infinity Cache when it is drawn into again
        rotate = 1
        if world == 1 and slots_used:
            rotate = max(1, -(-int(args.attachment_gb * 1e9) // (fif * fpl * 12 * npix)))
        slot_planes = [torch.empty(3, npix, dtype=torch.int32, device="cuda") for _ in range(fif * fpl * rotate)]
        slot_ptr = [[p[i].data_ptr() for i in range(3)] for p in slot_planes]
        planes, ptr = slot_planes[0], slot_ptr[0]
        if world > 1 and rank == 0:
            gathered = torch.empty(world, 3, npix, dtype=torch.int32, device="cuda")
            frame = torch.empty(3, h * w, dtype=torch.int32, device="cuda")
        prm0 = R.make_params(w, h, tile=tile, nranks=world, flags=sflags & R.ROOT_IDLE)

    # Render-kernel duration: with one GPU two HIP events per stream bracket that stream's launches of the timed
    # region (a stream holds nothing else, so their distance / launches is the average launch-to-launch time of
    # the kernel on it; a pair of events around every launch would add about 3 us of marker latency).  With several
    # slots that is the duration of a launch that shares the GPU with the other slots' launches.  With several
    # GPUs the stream also carries the gather, so there a pair of events brackets every 4th render launch.
    every = 4
    nev = (nframes + every - 1) // every
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(nev)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(nev)]
    ev_begin, ev_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    slot_streams = slot_ev = submissions = None
    if slots_used:
        slot_streams = [torch.cuda.ExternalStream(ctx.frame_stream(j)) for j in range(fif)]
        slot_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(fif)]

        # Submission k draws frames k*fpl .. on slot k % fif into that slot's attachment sets; the argument arrays
        # are built once (the host work of a draw -- camera inverse, hit rectangle, grid -- stays in the call).
        drawn = set()  # attachment sets some submission draws into
        last_set = [0]  # the set the last frame of the latest list of submissions goes to

        def make_submissions(nframes, rotate):
            # with rotate = M every slot goes round M groups of fpl sets, so that M * fif * fpl sets are in use
            ptrs = slot_ptr
            subs, k, i = [], 0, 0
            while i < nframes:
                n = min(fpl, nframes - i)
                j = k % fif
                group = j + fif * ((k // fif) % rotate)
                subs.append((j, n, scene.make_batch([cams[(i + u) % len(cams)] if turntable else cam for u in range(n)],
                                                    [prm] * n, [ptrs[group * fpl + u] for u in range(n)])))
                drawn.update(group * fpl + u for u in range(n))
                last_set[0] = group * fpl + n - 1
                i, k = i + n, k + 1
            return subs

        warm_submissions = make_submissions(max(nwarm, fif * fpl * rotate), rotate)  # every set is touched once
        submissions = make_submissions(nframes, rotate)
        final_set = last_set[0]

    def run(subs):
        for j, n, batch in subs:
            scene.render_batch(batch, j)

    # several ranks, grouped sparse composite: group g draws its frames into buffer set g % 2 and composites them
    def make_groups(nframes):
        return [(g % 2, min(fpl, nframes - g * fpl)) for g in range((nframes + fpl - 1) // fpl)]

    def composite_group(cb, k, n):
        if direct:
            cb.composite(k, n)
        else:  # rehearsal: the collective runs on host copies
            ctx.frames_join()
            if cb.slot_elems:
                host = RD.gather_to_rank0(cb.planes[k][:n].cpu(), rank, world)
                if rank == 0:
                    cb.gathered[k][:, :n].copy_(host)
            cb.scatter_received(k, n)

    def run_groups(groups):
        for k, n in groups:
            cb.render(k, n)
            composite_group(cb, k, n)

    if grouped:
        slot_streams = cb.slot_streams
        slot_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(2)]
        groups, warm_groups = make_groups(nframes), make_groups(nwarm)

    def step(i=None):
        timed = world > 1 and i is not None and i % every == 0
        if timed:
            ev0[i // every].record(stream)
        if sparse:
            sc.render()
        else:
            scene.render_device(cams[(i or 0) % len(cams)] if turntable else cam, prm, ptr[0], ptr[1], ptr[2])
        if timed:
            ev1[i // every].record(stream)
        if sparse:
            if direct:
                sc.composite()
            elif sc.slot_elems:  # rehearsal: the collective runs on host copies
                stream.synchronize()
                host = RD.gather_to_rank0(sc.planes.cpu(), rank, world)
                if rank == 0:
                    gathered.copy_(host)
                sc.scatter_received()
        elif world > 1:
            if args.backend == "nccl":
                RD.gather_to_rank0(planes, rank, world, gathered)
            else:  # rehearsal: the collective runs on host copies
                stream.synchronize()
                host = RD.gather_to_rank0(planes.cpu(), rank, world)
                if rank == 0:
                    gathered.copy_(host)
            if rank == 0:
                RD.deinterleave_on_gpu(ctx, prm0, gathered, frame)

    def warm_up():
        if grouped:
            run_groups(warm_groups)
        elif slots_used:
            run(warm_submissions)
        else:
            for _ in range(nwarm):
                step()
        torch.cuda.synchronize()

    # --warmup steps, then more of the same until --warmup-ms have passed (every rank the same number of rounds)
    t_warm, warm_rounds = time.perf_counter(), 0
    while True:
        warm_up()
        warm_rounds += 1
        go_on = torch.tensor([1.0 if (time.perf_counter() - t_warm) * 1e3 < args.warmup_ms else 0.0], device="cuda")
        if world > 1:
            dist.all_reduce(go_on, op=dist.ReduceOp.MAX)
        if float(go_on[0]) == 0.0 or warm_rounds >= 10000:
            break
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev_begin.record(stream)
    if grouped:
        for j in range(2):
            slot_ev[j][0].record(slot_streams[j])
        run_groups(groups)
        for j in range(2):
            slot_ev[j][1].record(slot_streams[j])
        ctx.frames_join()
    elif slots_used:
        for j in range(fif):
            slot_ev[j][0].record(slot_streams[j])
        run(submissions)
        for j in range(fif):
            slot_ev[j][1].record(slot_streams[j])
        ctx.frames_join()  # the context's stream (and ev_end on it) waits for every slot
    else:
        for i in range(nframes):
            step(i)
    ev_end.record(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if grouped:  # per submission of up to `fpl` frames: a slot's span (it also waits for the set's composite)
        counts = [sum(1 for k, n in groups if k == j) for j in range(2)]
        kernel_ms = sum(slot_ev[j][0].elapsed_time(slot_ev[j][1]) for j in range(2) if counts[j]) / max(1, sum(counts))
    elif world > 1:
        kernel_ms = sum(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / nev
    elif slots_used:  # per launch: a slot's span over the submissions it took
        counts = [sum(1 for sub in submissions if sub[0] == j) for j in range(fif)]
        kernel_ms = sum(slot_ev[j][0].elapsed_time(slot_ev[j][1]) for j in range(fif) if counts[j]) / max(1, sum(counts))
    else:
        kernel_ms = ev_begin.elapsed_time(ev_end) / nframes

    alone_ms = None
    if slots_used and not args.no_extra_passes:
        # after the timed region: the same submissions on one slot only, i.e. the launches with the GPU to themselves
        # (the last frame lands where the timed region left it)
        last_slot = submissions[-1][0]
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record(slot_streams[last_slot])
        for j, n, batch in submissions:
            scene.render_batch(batch, last_slot)
        a1.record(slot_streams[last_slot])
        torch.cuda.synchronize()
        alone_ms = a0.elapsed_time(a1) / len(submissions)

    # Frames drawn one lane per triangle never touch the LBVH: the same frames through it (the tile kernel behind a
    # binning pass, INV_TILE_TRAVERSAL), timed the same way on the same slots, for comparison
    lbvh_path_ms = None
    if world == 1 and slots_used and not args.no_extra_passes and not turntable:
        path_now = scene.draw_path(cam, prm)
        if (path_now & 15) == R.api.PATH_SCATTER:
            tprm = R.make_params(w, h, spp=spp, flags=flags | R.TILE_TRAVERSAL, ao_rays=ao_rays)
            n_t = min(len(submissions), 8)
            tsubs = [(j, n, scene.make_batch([cam] * n, [tprm] * n, [slot_ptr[(k * fpl + u) % len(slot_ptr)] for u in range(n)]))
                     for k, (j, n, _) in enumerate(submissions[:n_t])]
            run(tsubs)
            torch.cuda.synchronize()
            ts = time.perf_counter()
            run(tsubs)
            torch.cuda.synchronize()
            lbvh_path_ms = (time.perf_counter() - ts) / sum(n for _, n, _ in tsubs) * 1e3
            run(submissions[-min(len(submissions), fif * rotate):])  # the attachments hold the timed path's frames again
            torch.cuda.synchronize()

    # The same job through one group of sets per slot only (fif * fpl sets: 299 MB for the headline, less than the
    # Infinity Cache could hold if the stores were kept there): the stores are non-temporal, so this should make no
    # difference; reported to show that the timed region does not lean on that cache.
    resident = None
    if slots_used and rotate > 1 and not turntable and not args.no_extra_passes:
        subs = make_submissions(nframes, 1)
        run(make_submissions(fif * fpl, 1))
        torch.cuda.synchronize()
        ts = time.perf_counter()
        run(subs)
        torch.cuda.synchronize()
        resident = (time.perf_counter() - ts, fif * fpl)
    if slots_used and not turntable and not all(torch.equal(slot_planes[0], slot_planes[i]) for i in sorted(drawn)):
        raise SystemExit("the attachment sets hold different frames: timing rejected")

    t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, kernel_ms = float(t[0]), float(t[1])

    if world > 1 and rank == 0:
        # the composite must equal the same frame rendered by this rank alone
        alone = torch.empty(3, h * w, dtype=torch.int32, device="cuda")
        scene.render_device(cam, R.make_params(w, h, spp=spp, flags=flags, ao_rays=ao_rays), alone[0].data_ptr(), alone[1].data_ptr(),
                            alone[2].data_ptr())
        stream.synchronize()
        if grouped:  # every frame of the last group drawn into each set
            last = {}
            for k, n in warm_groups + groups:
                last[k] = n
            for k, n in last.items():
                for j in range(n):
                    if not torch.equal(alone, cb.frames[k][j]):
                        raise SystemExit("composited frame %d of set %d differs from the single-GPU frame: timing rejected" % (j, k))
            frame = cb.frames[groups[-1][0]][0] if groups else alone
        elif not torch.equal(alone, frame):
            raise SystemExit("composited frame differs from the single-GPU frame: timing rejected")

    if world == 1 and slots_used:
        planes = slot_planes[final_set]  # the attachments of the timed region's last frame
    if rank == 0:
        final = frame if world > 1 else planes
        hit_pixels = int((final[2] != -1).sum()) if world == 1 or rank == 0 else 0
        rays_per_frame = w * h * spp + ao_rays * hit_pixels  # primary + occlusion rays of the whole frame
        value = rays_per_frame * nframes / elapsed / 1e6
        # HBM roofline of the render kernel: every pixel's colour, depth and primitive id written once,
        # triangles (48 B) and LBVH nodes (64 B) read once per frame per GPU.
        # (frames drawn one lane per triangle read the triangle records, 48 B each, and never touch the LBVH's nodes)
        path = scene.draw_path(cam, R.make_params(w, h, spp=spp, flags=flags, ao_rays=ao_rays))
        scatter = (path & 15) == R.api.PATH_SCATTER
        read_bytes = ntri * 48 if scatter else scene_bytes
        alg_bytes = 12 * (npix if world > 1 else w * h) + read_bytes  # per frame
        launch_bytes = alg_bytes * fpl                                 # per launch of the render kernel
        achieved = launch_bytes / (kernel_ms * 1e-3) / 1e9
        kernel_name = ("k_render_brute" if args.brute_force else "k_scatter_triangles+k_scatter_resolve4" if scatter
                       else "k_render_tile_batch" if fpl > 1 and not ao_rays and spp < 4
                       else "k_render_tile") + ("+k_ao" if ao_rays else "")
        path_names = {R.api.PATH_BRUTE: "brute force", R.api.PATH_LANES: "one lane per ray", R.api.PATH_BLOCKS: "16x16 blocks",
                      R.api.PATH_TILES: "8x8 tiles", R.api.PATH_TILES_BATCHED: "8x8 tiles in batches",
                      R.api.PATH_SCATTER: "one lane per triangle + resolve"}
        result = {
            "metric": "Mrays/s (primary + AO) bunny @1080p, 1/2/4/8 MI355X; % HBM/VALU roofline",
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "mesh": "bunny (2503 v / 4968 f, regenerated from the fixture)",
                       "width": w, "height": h, "spp": spp, "ao_rays": ao_rays, "hit_pixels": hit_pixels,
                       # 98.8 % of the primary rays miss the scene's bounds: the rate over rays of hit pixels only
                       "hit_pixel_mrays_per_s": (hit_pixels * spp + ao_rays * hit_pixels) * nframes / elapsed / 1e6,
                       "rays_per_frame": rays_per_frame, "frames_per_step": args.frames_per_step,
                       "us_per_frame": elapsed / nframes * 1e6, "flags": int(flags),
                       "triangles": ntri, "lbvh_nodes": nnodes, "instances": side * side,
                       # how the timed frames were traced: frames that share a launch go through 16x16 blocks, a frame
                       # drawn on its own the way inv_draw_path says
                       "draw_path": ("16x16 blocks, %d frames per launch" % fpl if kernel_name.startswith("k_render_tile_batch")
                                     else path_names[path & 15] + (" after a binning pass" if path & R.api.PATH_BINNED else "")),
                       "bytes_read_per_frame": read_bytes,
                       # ms per frame of the same frames drawn through the LBVH (tile kernel behind a binning pass) when the
                       # timed frames were drawn one lane per triangle, which never touches the tree; else null
                       "lbvh_path_ms": lbvh_path_ms,
                       # host wall of the scene build (upload from a pinned mesh, assembly, LBVH): the first build (with allocations)
                       # and the median of the five that follow
                       "build_ms": [build_ms[0], float(np.median(build_ms[1:]))], "mesh_pinned": True, "tile": tile, "frames_in_flight": fif, "frames_per_launch": fpl,
                       "parallelism": ("%s tiles %% %d + RCCL gather to rank 0%s" % (
                                           args.composite, world, (", rank 0 draws no tiles" if idle else "") + (", %d frames per gather, 2 buffer sets" % fpl if grouped else "")
                                           + (", exchange through inv_comm_gather" if comm is not None else ""))
                                       if world > 1 else "1 GPU")},
            # `achieved` = algorithmic bytes of the frames drawn / wall time of the timed region: launches of different
            # frame slots overlap on the GPU, so the job's rate, not a launch's, is what the HBM interface sustained.
            # Every frame goes to attachments that left the Infinity Cache long ago (config.attachment_mb).
            "roofline": {"bound": "hbm", "achieved": alg_bytes * nframes / elapsed / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg_bytes * nframes / elapsed / 1e9 / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(args.workload, world, kernel_name.split("+")[0], fpl),
                         "kernel": kernel_name, "algorithmic_bytes": launch_bytes, "frames_per_launch": fpl,
                         # one launch of the kernel: a slot's span between two HIP events on its stream / its launches;
                         # with several slots a launch shares the GPU with the other slots' launches
                         "launch": {"kernel_ms": kernel_ms, "concurrent_launches": fif if slots_used else 1,
                                    "achieved": achieved, "frac": achieved / HBM_PEAK_GBS},
                         # `frac` is the job's (launches of several slots overlap); one launch on its own reaches
                         # per_launch_frac (the replay of the same submissions on one slot, roofline.alone)
                         "per_launch_frac": None},
            "cpu_baseline": None,
        }
        set_mb = 12 * (w * h) / 1e6
        if world == 1:
            result["config"]["attachment_sets"] = len(slot_planes)
            result["config"]["attachment_mb"] = set_mb * len(slot_planes)  # against the 256 MB Infinity Cache
        if resident is not None:  # the same job through fif * fpl sets only: not an HBM figure if it differed
            st, nsets = resident
            result["roofline"]["infinity_cache_sized_sets"] = {
                "attachment_sets": nsets, "attachment_mb": set_mb * nsets, "us_per_frame": st / nframes * 1e6,
                "value": rays_per_frame * nframes / st / 1e6, "achieved": alg_bytes * nframes / st / 1e9,
                "frac_of_hbm_peak": alg_bytes * nframes / st / 1e9 / HBM_PEAK_GBS}
        if alone_ms is not None:  # the same launches when no other slot's launch shares the GPU with them
            result["roofline"]["alone"] = {"kernel_ms": alone_ms, "achieved": launch_bytes / (alone_ms * 1e-3) / 1e9,
                                           "frac": launch_bytes / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            result["roofline"]["per_launch_frac"] = result["roofline"]["alone"]["frac"]
        else:
            result["roofline"]["per_launch_frac"] = achieved / HBM_PEAK_GBS
        # VALU side: utilisation from the newest committed rocprofv3 counter pass of this workload (the counters
        # cannot be read inside this process), or None
        result["roofline"]["valu"] = pmc_valu(args.workload, world, kernel_name.split("+")[-1])
        valu_bound = bool(ao_rays) or spp >= 4  # occlusion rays and many samples per pixel: vector arithmetic, not stores

    # Several ranks: the headline frame is bound by rank 0's own stores (every pixel of every frame is written into
    # rank 0's HBM whatever N is), so sharding cannot pay there; these short passes (not part of `value`) time the
    # timed workload and the two heavier ones the same way, as a stream of frames in groups of `xf` per submission
    # and per gather on two buffer sets: the sharded draws alone, the draws plus the composite, and the whole frames
    # on one GPU (rank 0 alone on two frame slots, the others idle); per frame, max over ranks.
    extra, scaling_workloads = {}, {}
    if world > 1 and sparse and not args.no_extra_workloads:
        # These passes come after the timed region and must not cost its result: if a rank is still in them after
        # --extras-watchdog-seconds, rank 0 prints the line it has, with a note, and every rank ends.
        def extras_give_up():
            sys.stderr.write("bench.py: rank %d still in the passes over the heavier workloads after %g s; ending without them\n"
                             % (rank, args.extras_watchdog_seconds))
            if rank == 0:
                result["extra_workloads_error"] = ("a rank was still in the passes over the heavier workloads after %g s; "
                                                   "they were given up, the timed region before them was complete"
                                                   % args.extras_watchdog_seconds)
                if extra:
                    result["extra_workloads"] = extra
                    result["scaling_workloads"] = scaling_workloads
                print(json.dumps(result), flush=True)
            sys.stderr.flush()
            os._exit(0)

        extras_watchdog = None
        if args.extras_watchdog_seconds > 0:
            import threading
            extras_watchdog = threading.Timer(args.extras_watchdog_seconds, extras_give_up)
            extras_watchdog.daemon = True
            extras_watchdog.start()
        rehearsal = args.backend != "nccl"  # ranks sharing one GPU: enough frames to see that it works

        def timed(body, n, frames_each):
            torch.cuda.synchronize()
            dist.barrier()
            t_0 = time.perf_counter()
            for i in range(n):
                body(i)
            ctx.frames_join()
            torch.cuda.synchronize()
            tt = torch.tensor([time.perf_counter() - t_0], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt[0]) / (n * frames_each) * 1e3

        trace_on = bool(os.environ.get("INV_BENCH_TRACE"))

        def trace(*what):  # where a rank is, for a run that has to be looked into
            if trace_on:
                print("bench.py trace: rank %d" % rank, *what, file=sys.stderr, flush=True)

        names = [args.workload] + [n_ for n_ in ("bunny_4k_16spp", "bunny_1080p_ao64_closeup") if n_ != args.workload]
        for name in names:
            trace(name, "begin")
            xw = WORKLOADS[name]
            if xw.get("instances_side", 1) != side or xw.get("turntable"):
                continue  # another scene than the one built
            ew, eh = xw["width"], xw["height"]
            ecam = R.camera_reference(ew, eh)
            if side > 1:
                ecam = common.grid_camera(ecam, side)
            if "view_translation" in xw:
                ecam.view[12], ecam.view[13], ecam.view[14] = xw["view_translation"]
            kw = dict(spp=xw["spp"], flags=xw["flags"], ao_rays=xw.get("ao_rays", 0))
            heavy = bool(kw["ao_rays"]) or ew * eh * kw["spp"] > 1 << 26
            xf = (2 if rehearsal else 4) if heavy else (4 if rehearsal else 8)  # frames per group
            n_x = 2 if rehearsal else (6 if heavy else 24)                      # groups timed
            # the whole frames on one GPU: rank 0 alone, groups of xf frames on two frame slots in turn
            eprm = R.make_params(ew, eh, **kw)
            one_ms, one = None, None
            if rank == 0:
                one = torch.empty(2, xf, 3, eh * ew, dtype=torch.int32, device="cuda")
                one_batches = [scene.make_batch([ecam] * xf, [eprm] * xf,
                                                [tuple(one[j, u, p_].data_ptr() for p_ in range(3)) for u in range(xf)])
                               for j in range(2)]

            # (Ranks that share one GPU, occlusion rays: on one slot.  With a second process on the device, rank 0's
            # occlusion frames on two slots at once have now and then never completed -- one run in seven, only in
            # this pass, never with one process per GPU or with one process alone, 40 fresh processes and 900 frames
            # in tools/ao_two_slots_probe.py -- and a rehearsal's times mean nothing anyway.)
            one_slot = rehearsal and bool(kw["ao_rays"])

            def alone(i):
                if rank == 0:
                    scene.render_batch(one_batches[i % 2], 0 if one_slot else i % 2)

            alone(0)
            alone(1)
            trace(name, "one GPU: warmed")
            one_ms = timed(alone, n_x, xf)
            trace(name, "one GPU: timed")
            for share, share_flag in [("equal", 0)] + ([("rank0_idle", R.ROOT_IDLE)] if world >= 3 else []):
                ecb = RD.SparseCompositeBatch(scene, ecam, ew, eh, rank, world, frames=xf, sets=2, tile=32, comm=comm,
                                              **dict(kw, flags=kw["flags"] | share_flag))

                def draws(i):
                    ecb.render(i % 2, xf)

                def both(i):
                    ecb.render(i % 2, xf)
                    composite_group(ecb, i % 2, xf)

                trace(name, share, "composite made")
                for i in range(2):
                    both(i)
                render_ms = timed(draws, n_x, xf)
                trace(name, share, "draws timed")
                both_ms = timed(both, n_x, xf)
                trace(name, share, "draws + composite timed")
                both(0)
                ctx.frames_join()
                torch.cuda.synchronize()
                ok = all(bool(torch.equal(one[0, 0], ecb.frames[0][u])) for u in range(xf)) if rank == 0 else True
                res = {"render_only_ms": render_ms, "render_plus_gather_ms": both_ms, "one_gpu_ms": one_ms,
                       "composite_equals_one_gpu_frame": ok, "frames_timed": n_x * xf, "frames_per_gather": xf}
                if rank == 0 and not ok:
                    raise SystemExit("extra workload %s (%s): the composite differs from the single-GPU frame" % (name, share))
                if share == "equal":
                    extra[name] = res
                else:  # the same with rank 0 drawing no tiles
                    extra[name][share] = {k: res[k] for k in ("render_only_ms", "render_plus_gather_ms", "composite_equals_one_gpu_frame")}
                del ecb
            del one
            # the summary the scaling curve is read from: per frame, the better of the two shares of rank 0
            best = min([("equal", extra[name])] + ([("none", extra[name]["rank0_idle"])] if "rank0_idle" in extra[name] else []),
                       key=lambda sv: sv[1]["render_plus_gather_ms"])
            scaling_workloads[name] = {
                "one_gpu_ms": one_ms, "render_only_ms": best[1]["render_only_ms"],
                "render_plus_gather_ms": best[1]["render_plus_gather_ms"], "root_share": best[0], "frames_per_gather": xf,
                "speedup_over_one_gpu": one_ms / best[1]["render_plus_gather_ms"], "n_gpus": world}

        if extras_watchdog is not None:
            extras_watchdog.cancel()

    if rank == 0:
        if extra:
            result["extra_workloads"] = extra
            result["scaling_workloads"] = scaling_workloads
        if world == 1 and not args.no_cpu_baseline:
            # (turntable: the frame left in the buffers is the last step's)
            base, oracle_out, work = cpu_baseline(wl, cams[(nframes - 1) % len(cams)] if turntable else cam,
                                                  budget_s=args.cpu_baseline_seconds)
            # outputs are compared before the timing is accepted
            got = planes.cpu().numpy()
            ok = (np.array_equal(got[2].view(np.uint32).reshape(h, w), oracle_out[2])
                  and np.array_equal(got[1].view(np.uint32).reshape(h, w), oracle_out[1].view(np.uint32))
                  and np.array_equal(got[0].view(np.uint8).reshape(h, w, 4), oracle_out[0]))
            if not ok:
                raise SystemExit("GPU output differs from the CPU oracle: timing rejected")
            base["outputs_match_gpu"] = True
            result["cpu_baseline"] = base
            # not a utilisation: the arithmetic a per-ray LBVH traversal of this frame would do (the oracle's test
            # counts: 20 flops per ray, 22 per box test, 51 per triangle test, 15 per shaded hit), for comparison with
            # the instruction counts of the kernel's counter pass
            result["cpu_baseline"]["oracle_traversal_work"] = work
        if valu_bound:
            # The frame is bound by vector arithmetic: the roofline is the fp32 vector peak.  Instructions cannot be
            # counted inside this process, so `achieved` = the vector instructions the newest committed counter pass of this
            # workload saw per launch of each of the frame's kernels (times the lanes active, 2 flops each as if all were
            # multiply-adds) over THIS run's time per frame; the HBM view of the same run is kept beside it.  Without
            # such a pass the line keeps the HBM view.
            r = result["roofline"]
            flops, used = 0.0, []
            for k in ("k_render_tile", "k_ao"):
                v = pmc_valu(args.workload, world, k)
                if v and v.get("valu_instructions_per_launch"):
                    flops += v["valu_instructions_per_launch"] * v["lanes_active"] * 128.0
                    used.append("%s (%s)" % (k, v["source"]))
            if flops:
                hbm_view = {k: r[k] for k in ("achieved", "peak", "unit", "frac", "algorithmic_bytes", "traffic")}
                tf = flops * nframes / elapsed / 1e12
                r.update({"bound": "valu", "achieved": tf, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / VALU_PEAK_TFLOPS,
                          "flops_per_frame": flops, "flops_basis": "vector instructions x active lanes x 2 of " + ", ".join(used),
                          "per_launch_frac": None, "hbm": hbm_view})
        print(json.dumps(result))
    if comm is not None:
        comm.close()
    scene.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
