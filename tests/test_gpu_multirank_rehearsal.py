"""The N > 1 code path of bench.py on one GPU: two and three ranks share cuda:0 and gather through gloo.

bench.py checks the composited frame against rank 0's own full render before it prints its line, so a zero exit
status means the tile dealing, the per-rank draws, the gather and the re-assembly agree with the single-GPU frame.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(nranks, *extra, steps=3, warmup=1):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", str(nranks), "--steps", str(steps), "--warmup", str(warmup), "--frames-per-step", "1",
           "--backend", "gloo", "--watchdog-seconds", "120", *extra]
    # (the watchdog ends a rank that hangs, with its stacks, before this test's own limit would leave the ranks behind)
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=170)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.timeout(180)
@pytest.mark.parametrize("nranks,extra", [(2, ()), (3, ("--workload", "bunny_1080p_ao64")), (2, ("--composite", "dense")),
                                          (3, ("--root-share", "none")), (4, ("--composite-frames", "1")),
                                          (3, ("--composite", "dense", "--root-share", "none"))])
def test_bench_with_several_ranks_on_one_gpu(nranks, extra):
    r = _run(nranks, *extra)
    assert r["n_gpus"] == nranks and r["steps"] == 3 and r["scaling"] == "strong"
    assert r["value"] > 0 and r["config"]["hit_pixels"] == 25139
    assert ("dense" if "dense" in extra else "sparse") in r["config"]["parallelism"]
    # rank 0 draws no tiles when asked to, and by default from four ranks on
    assert ("rank 0 draws no tiles" in r["config"]["parallelism"]) == ("none" in extra or nranks >= 4)
    if nranks >= 3 and "dense" not in extra:  # the heavier workloads are timed with both shares
        for x in r["extra_workloads"].values():
            assert x["composite_equals_one_gpu_frame"] and x["rank0_idle"]["composite_equals_one_gpu_frame"]


@pytest.mark.gpu
@pytest.mark.timeout(180)
def test_bench_keeps_the_timed_result_when_the_passes_after_it_are_given_up():
    """The passes over the heavier workloads follow the timed region; a rank that is still in them when their own
    time limit ends (here: at once) must not cost the line: it is printed with a note and every rank ends well."""
    r = _run(2, "--extras-watchdog-seconds", "0.02")
    assert r["n_gpus"] == 2 and r["value"] > 0 and r["config"]["hit_pixels"] == 25139
    assert "given up" in r["extra_workloads_error"]


@pytest.mark.gpu
@pytest.mark.timeout(180)
@pytest.mark.parametrize("nranks,frames,steps,warmup", [(2, 4, 19, 9), (3, 8, 20, 3), (2, 1, 3, 1)])
def test_bench_grouped_composite_on_one_gpu(nranks, frames, steps, warmup):
    """Groups of `frames` frames per submission and gather, alternating between two buffer sets, with short last
    groups (steps and warm-up are not multiples of the group size); frames = 1 is the frame-by-frame composite."""
    r = _run(nranks, "--composite-frames", str(frames), steps=steps, warmup=warmup)
    assert r["n_gpus"] == nranks and r["steps"] == steps and r["config"]["hit_pixels"] == 25139
    assert (("%d frames per gather" % frames) in r["config"]["parallelism"]) == (frames > 1)


@pytest.mark.gpu
@pytest.mark.timeout(180)
@pytest.mark.parametrize("nranks,extra", [(2, ("--composite-frames", "1")), (3, ("--composite-frames", "4"))])
def test_bench_with_the_librarys_own_exchange(nranks, extra):
    """--gather c: the tiles travel through inv_comm_gather (here its staging files; RCCL send / receive with the
    nccl backend) instead of torch.distributed.gather, frame by frame and in groups of four."""
    r = _run(nranks, "--gather", "c", *extra, steps=5, warmup=2)
    assert r["n_gpus"] == nranks and r["config"]["hit_pixels"] == 25139
    assert "inv_comm_gather" in r["config"]["parallelism"]


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


@pytest.mark.gpu
@pytest.mark.timeout(200)
@pytest.mark.parametrize("nranks,launcher,extra", [(2, "torchrun", ()), (3, "spawn", ("--no-extra-workloads",))])
def test_bench_starts_its_own_ranks(nranks, launcher, extra):
    """`python bench.py --gpus N` without torch.distributed.run and without WORLD_SIZE: the parent starts the N ranks
    itself (before anything opens the GPU; through torch's launcher, or directly), exactly one line comes out, and it
    carries the per-workload phase times the scaling curve is read from."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(nranks), "--backend", "gloo", "--steps", "3",
           "--warmup", "1", "--frames-per-step", "2", "--watchdog-seconds", "100", *extra]
    p = subprocess.run(cmd, cwd=ROOT, env=_clean_env(INV_BENCH_LAUNCHER=launcher), capture_output=True, text=True, timeout=190)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]  # what gloo prints on standard output was sent to standard error
    r = json.loads(lines[0])
    assert r["n_gpus"] == nranks and r["steps"] == 3 and r["value"] > 0
    if extra:
        assert "scaling_workloads" not in r
        return
    sw = r["scaling_workloads"]
    assert set(sw) == {"bunny_1080p_primary_lambert_invert", "bunny_4k_16spp", "bunny_1080p_ao64_closeup"}
    for v in sw.values():
        assert v["one_gpu_ms"] > 0 and v["render_only_ms"] > 0 and v["render_plus_gather_ms"] > 0 and v["n_gpus"] == nranks


@pytest.mark.gpu
@pytest.mark.timeout(120)
def test_bench_launcher_reports_a_failing_rank():
    """A rank that cannot start (more RCCL ranks than GPUs on this box) ends the whole launch with a non-zero status
    and no JSON line, instead of leaving the others waiting."""
    import torch
    n = torch.cuda.device_count() + 1
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       cwd=ROOT, env=_clean_env(), capture_output=True, text=True, timeout=110)
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert "needs %d GPUs" % n in p.stderr
