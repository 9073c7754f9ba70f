"""CPU tests of the multi-GPU path with the gloo backend, world sizes 2 and 3.

What shards is the frame: rank r owns the screen tiles with tile_id % world == r and stores them tile after tile.
These tests run the product's tile layout (inv_tile_layout), the gather to rank 0 (renderer_amd.dist) and the
re-assembly on a frame whose pixels come from the committed golden, with host tensors instead of HBM.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import common
import renderer_amd as R
from renderer_amd import dist as RD


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _slots_from_frame(frame, prm):
    """What rank `prm.rank` would render: its tiles cut out of a full frame, partial edge tiles zero-padded."""
    h, w = frame.shape
    t = prm.tile
    npix, ids = R.tile_layout(prm)
    tiles_x = (w + t - 1) // t
    slots = np.zeros((npix // (t * t), t, t), frame.dtype)
    for slot, tid in enumerate(ids):
        ty, tx = divmod(int(tid), tiles_x)
        part = frame[ty * t:(ty + 1) * t, tx * t:(tx + 1) * t]
        slots[slot, :part.shape[0], :part.shape[1]] = part
    return slots


def _worker(rank, world, port, w, h, tile, result_path):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    color, depth, primid = common.golden_frame("bunny_512")
    full = [np.ascontiguousarray(color).view(np.uint32).reshape(512, 512)[:h, :w],
            depth.view(np.uint32)[:h, :w], primid[:h, :w]]
    prm = RD.shard_params(w, h, rank, world, tile=tile)
    planes = RD.alloc_planes(prm, "cpu")
    for k in range(3):
        planes[k] = torch.from_numpy(_slots_from_frame(full[k], prm).reshape(-1).view(np.int32))
    gathered = RD.gather_to_rank0(planes, rank, world)
    ok = True
    if rank == 0:
        g = gathered.numpy().view(np.uint32)  # [world, 3, npix]
        for k in range(3):
            per_rank = [g[r, k].reshape(-1, tile, tile) for r in range(world)]
            frame = common.slots_to_frame(per_rank, w, h, tile, world)
            ok = ok and np.array_equal(frame, full[k])
        with open(result_path, "w") as f:
            f.write("ok" if ok else "mismatch")
    else:
        assert gathered is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h,tile", [(2, 512, 512, 64), (3, 500, 300, 32), (2, 96, 40, 64)])
def test_tiles_gather_to_rank0(tmp_path, world, w, h, tile):
    result = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(world, _free_port(), w, h, tile, result), nprocs=world, join=True)
    assert open(result).read() == "ok"


def test_every_tile_has_exactly_one_owner():
    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            npix, ids = R.tile_layout(RD.shard_params(1920, 1080, r, world))
            slots = -(-510 // world)
            assert npix == (1920 * 1080 if world == 1 else slots * 64 * 64)
            seen += [int(i) for i in ids]
        assert sorted(seen) == list(range(30 * 17))
    # the bunny's 207x197 pixel footprint at 1080p touches 16 tiles of 64x64: with 8 ranks no rank gets more than 3
    hit_tiles = [ty * 30 + tx for ty in range(301 // 64, 496 // 64 + 1) for tx in range(880 // 64, 1085 // 64 + 1)]
    per_rank = np.bincount([t % 8 for t in hit_tiles], minlength=8)
    assert per_rank.max() <= 3 and per_rank.min() >= 1
