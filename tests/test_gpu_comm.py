"""The exchange step of the multi-GPU composite through the C ABI (inv_comm_*), rehearsed on one GPU.

RCCL refuses two ranks on one device, so the several-rank runs use the host transport (staging files in a directory
under /dev/shm); the RCCL transport is opened, given an id and a one-rank communicator (the library is loaded with
dlopen, the symbols resolve, ncclCommInitRank runs).  Real several-GPU runs are the driver's scaling bench.
"""
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest
import torch

import renderer_amd as R

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tools", "headless", "invert_headless")


def _read(prefix, w, h):
    raw = open(prefix + ".ppm", "rb").read()
    rgb = np.frombuffer(raw[len(("P6\n%d %d\n255\n" % (w, h)).encode()):], np.uint8).reshape(h, w, 3)
    raw = open(prefix + ".pfm", "rb").read()
    depth = np.frombuffer(raw[len(("Pf\n%d %d\n-1.0\n" % (w, h)).encode()):], "<f4").reshape(h, w)[::-1]
    primid = np.frombuffer(open(prefix + ".pid", "rb").read()[16:], "<u4").reshape(h, w)
    return rgb, depth, primid


@pytest.mark.parametrize("ranks,extra", [(2, []), (3, ["--ao", "16"]), (4, ["--spp", "4", "--turntable"]),
                                         (4, ["--rank0-idle"])])
def test_headless_ranks_over_the_host_transport(bunny_path, tmp_path, ranks, extra):
    """N processes of the C++ host program share this GPU: each draws its tiles, inv_comm_gather carries the slot
    buffers to rank 0 through files, rank 0's frame equals the frame of a single process."""
    if not os.path.exists(EXE):
        pytest.skip("tools/headless/invert_headless is not built")
    w, h = 1280, 720
    common = [EXE, "--obj", bunny_path, "--width", str(w), "--height", str(h), "--lambert", "--invert", "--frames", "3", *extra]
    single = str(tmp_path / "single")
    alone = [a for a in common if a != "--rank0-idle"]  # (a flag of sharded frames only)
    p = subprocess.run(alone + ["--in-flight", "1", "--per-launch", "1", "--out", single], capture_output=True, text=True, timeout=100)
    assert p.returncode == 0, p.stdout + p.stderr
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as meet:
        out = str(tmp_path / "composite")
        procs = [subprocess.Popen(common + ["--ranks", str(ranks), "--rank", str(r), "--gpu", "0", "--rendezvous", meet,
                                            "--transport", "host", "--out", out],
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(ranks)]
        results = []
        for r, pr in enumerate(procs):
            so, se = pr.communicate(timeout=150)
            assert pr.returncode == 0, "rank %d: %s%s" % (r, so, se)
            results.append(json.loads(so.strip().splitlines()[-1]))
        assert os.listdir(meet) == []  # every staging file was consumed
    assert [r["rank"] for r in results] == list(range(ranks)) and all(r["transport"] == "host" for r in results)
    sharing = ranks - ("--rank0-idle" in extra)  # ranks that draw tiles
    assert results[0]["tiles_dealt"] > ranks and results[0]["slot_elements"] == -(-results[0]["tiles_dealt"] // sharing) * 32 * 32
    got, want = _read(out, w, h), _read(single, w, h)
    for g, x in zip(got, want):
        assert np.array_equal(g.view(np.uint8), x.view(np.uint8))
    assert results[0]["hit_pixels_last_frame"] == int((want[2] != 0xFFFFFFFF).sum()) > 5000


def test_comm_gather_from_python_threads(bunny_path, tmp_path):
    """inv_comm_gather with three 'ranks' of this process (a context each, the host transport): counts, strides and
    the order of repeated exchanges."""
    import threading
    n, count, stride = 3, 1000, 1536
    ctxs = [R.Context(0) for _ in range(n)]
    comms = [R.api.Comm.host(ctxs[r], str(tmp_path), r, n) for r in range(n)]
    send = [torch.arange(count, dtype=torch.int32, device="cuda") + 100000 * r for r in range(n)]
    recv = torch.full((n * stride,), -7, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    errors = []

    def rank_main(r):
        try:
            for it in range(3):  # repeated exchanges keep their order
                send[r].add_(1)
                torch.cuda.synchronize()
                comms[r].gather(send[r].data_ptr(), count, recv.data_ptr() if r == 0 else 0, stride if r == 0 else 0)
                if r == 0:
                    ctxs[0].synchronize()
                    got = recv.cpu().numpy().reshape(n, stride)
                    for p in range(1, n):
                        assert np.array_equal(got[p, :count], np.arange(count) + 100000 * p + it + 1), (it, p)
                        assert (got[p, count:] == -7).all()
                    assert (got[0] == -7).all()  # rank 0's own part is not touched
        except Exception as e:  # noqa: BLE001
            errors.append((r, repr(e)))

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(n)]
    [t.start() for t in threads]
    [t.join(timeout=120) for t in threads]
    assert not errors, errors
    for c in comms:
        c.close()
    for c in ctxs:
        c.close()
    # argument checks
    ctx = R.Context(0)
    with pytest.raises(R.InvertError):
        R.api.Comm.host(ctx, str(tmp_path / "missing"), 0, 2)
    with pytest.raises(R.InvertError):
        R.api.Comm.host(ctx, str(tmp_path), 2, 2)
    ctx.close()


def test_rccl_transport_opens_and_creates_a_communicator():
    """librccl is found at run time, an id is made, and a communicator of one rank is created on the context's GPU and
    stream; a gather of one rank moves nothing."""
    ctx = R.Context(0)
    uid = R.api.comm_unique_id()
    assert len(uid) == R.api.COMM_ID_BYTES and any(uid)
    comm = R.api.Comm.rccl(ctx, uid, 0, 1)
    buf = torch.zeros(16, dtype=torch.int32, device="cuda")
    comm.gather(buf.data_ptr(), 16, buf.data_ptr(), 16)
    ctx.synchronize()
    comm.close()
    ctx.close()
