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


# ---- sparse composite: only the tiles under the scene's projected bounds travel ------------------------------

BUNNY_RECT_512 = (200, 130, 320, 260)  # a rectangle around the bunny's footprint in the 512x512 golden


def _sparse_worker(rank, world, port, w, h, tile, rect, result_path):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    color, depth, primid = common.golden_frame("bunny_512")
    full = [np.ascontiguousarray(color).view(np.uint32).reshape(512, 512)[:h, :w],
            depth.view(np.uint32)[:h, :w], primid[:h, :w]]
    prm = R.make_params(w, h, tile=tile, rank=rank, nranks=world, flags=R.SPARSE)
    grid, npix, ids = R.sparse_layout(prm, rect)
    _, slot_elems, _ = R.sparse_layout(R.make_params(w, h, tile=tile, rank=1, nranks=world, flags=R.SPARSE), rect)
    tiles_x = (w + tile - 1) // tile
    planes = torch.zeros(3, slot_elems, dtype=torch.int32)
    if rank > 0:  # what this rank would have rendered: its tiles cut out of the full frame
        assert npix == slot_elems
        for k in range(3):
            slots = np.zeros((slot_elems // (tile * tile), tile, tile), np.uint32)
            for slot, tid in enumerate(ids):
                ty, tx = divmod(int(tid), tiles_x)
                part = full[k][ty * tile:(ty + 1) * tile, tx * tile:(tx + 1) * tile]
                slots[slot, :part.shape[0], :part.shape[1]] = part
            planes[k] = torch.from_numpy(slots.reshape(-1).view(np.int32))
    ok = True
    if rank == 0:
        assert npix == w * h
        gathered = torch.zeros(world, 3, slot_elems, dtype=torch.int32)
        dist.gather(planes, list(gathered.unbind(0)), dst=0)
        g = gathered.numpy().view(np.uint32)
        for k in range(3):
            frame = full[k].copy()
            # rank 0 renders everything but the other ranks' tiles: blank those, then fill them from the gather
            for j in range(grid[2] * grid[3]):
                r, slot = j % world, j // world
                if r == 0:
                    continue
                ty, tx = grid[1] + j // grid[2], grid[0] + j % grid[2]
                y1, x1 = min((ty + 1) * tile, h), min((tx + 1) * tile, w)
                frame[ty * tile:y1, tx * tile:x1] = 0xDEADBEEF
                frame[ty * tile:y1, tx * tile:x1] = g[r, k].reshape(-1, tile, tile)[slot][:y1 - ty * tile, :x1 - tx * tile]
            ok = ok and np.array_equal(frame, full[k])
        with open(result_path, "w") as f:
            f.write("ok" if ok else "mismatch")
    else:
        dist.gather(planes, None, dst=0)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h,tile", [(2, 512, 512, 64), (3, 500, 300, 32)])
def test_sparse_tiles_gather_to_rank0(tmp_path, world, w, h, tile):
    result = str(tmp_path / "result.txt")
    mp.spawn(_sparse_worker, args=(world, _free_port(), w, h, tile, BUNNY_RECT_512, result), nprocs=world, join=True)
    assert open(result).read() == "ok"


def _grouped_worker(rank, world, port, w, h, tile, rect, frames, used, result_path):
    """The grouped sparse composite's exchange (renderer_amd.dist.SparseCompositeBatch.gather): `frames` frames per
    buffer set, of which a short last group uses only the first `used`; one gather of the leading slices, and on rank 0
    plane k of frame j of rank r sits r * frames * 3 * n + (3 * j + k) * n elements into the receive buffer, which is
    what inv_deinterleave_sparse is given (rank_stride = frames * 3 * n, plane_stride = n, nplanes = 3 * used)."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    color, depth, primid = common.golden_frame("bunny_512")
    full = [np.ascontiguousarray(color).view(np.uint32).reshape(512, 512)[:h, :w],
            depth.view(np.uint32)[:h, :w], primid[:h, :w]]
    prm = R.make_params(w, h, tile=tile, rank=rank, nranks=world, flags=R.SPARSE)
    grid, npix, ids = R.sparse_layout(prm, rect)
    _, n, _ = R.sparse_layout(R.make_params(w, h, tile=tile, rank=1, nranks=world, flags=R.SPARSE), rect)
    tiles_x = (w + tile - 1) // tile
    planes = torch.zeros(frames, 3, n, dtype=torch.int32)
    if rank > 0:  # frame j of this rank: its tiles cut out of the full frame, every value XOR j
        for j in range(frames):
            for k in range(3):
                slots = np.zeros((n // (tile * tile), tile, tile), np.uint32)
                for slot, tid in enumerate(ids):
                    ty, tx = divmod(int(tid), tiles_x)
                    part = full[k][ty * tile:(ty + 1) * tile, tx * tile:(tx + 1) * tile] ^ np.uint32(j)
                    slots[slot, :part.shape[0], :part.shape[1]] = part
                planes[j, k] = torch.from_numpy(slots.reshape(-1).view(np.int32))
    if rank == 0:
        gathered = torch.full((world, frames, 3, n), 0x3C3C3C3C, dtype=torch.int32)
        dist.gather(planes[:used], [gathered[r, :used] for r in range(world)], dst=0)
        flat = gathered.numpy().view(np.uint32).reshape(-1)
        ok = True
        for j in range(used):
            for k in range(3):
                for t in range(grid[2] * grid[3]):
                    r, slot = t % world, t // world
                    if r == 0:
                        continue
                    ty, tx = grid[1] + t // grid[2], grid[0] + t % grid[2]
                    y1, x1 = min((ty + 1) * tile, h), min((tx + 1) * tile, w)
                    at = r * frames * 3 * n + (3 * j + k) * n + slot * tile * tile
                    got = flat[at:at + tile * tile].reshape(tile, tile)[:y1 - ty * tile, :x1 - tx * tile]
                    ok = ok and np.array_equal(got, full[k][ty * tile:y1, tx * tile:x1] ^ np.uint32(j))
        ok = ok and bool((gathered[1:, used:] == 0x3C3C3C3C).all())  # frames past the short group are not touched
        with open(result_path, "w") as f:
            f.write("ok" if ok else "mismatch")
    else:
        dist.gather(planes[:used], None, dst=0)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h,tile,frames,used", [(2, 512, 512, 64, 4, 4), (3, 500, 300, 32, 8, 3)])
def test_grouped_sparse_gather(tmp_path, world, w, h, tile, frames, used):
    result = str(tmp_path / "result.txt")
    mp.spawn(_grouped_worker, args=(world, _free_port(), w, h, tile, BUNNY_RECT_512, frames, used, result), nprocs=world,
             join=True)
    assert open(result).read() == "ok"


def test_sparse_layout_deals_each_grid_tile_once():
    w, h, tile = 1920, 1080, 64
    rect = (880, 301, 1086, 497)  # the bunny's footprint at 1080p
    for world in (2, 4, 8):
        seen, slot_elems = [], None
        for r in range(world):
            grid, npix, ids = R.sparse_layout(R.make_params(w, h, tile=tile, rank=r, nranks=world, flags=R.SPARSE), rect)
            assert grid == (13, 4, 4, 4)
            if r == 0:
                assert npix == w * h
            else:
                assert npix == -(-16 // world) * tile * tile
            seen += [int(i) for i in ids]
        want = sorted((4 + ty) * 30 + 13 + tx for ty in range(4) for tx in range(4))
        assert sorted(seen) == want
    # rank 0 idle: the same tiles dealt to ranks 1.., rank 0 keeps the frame and takes none
    for world in (3, 4, 8):
        seen = []
        for r in range(world):
            prm = R.make_params(w, h, tile=tile, rank=r, nranks=world, flags=R.SPARSE | R.ROOT_IDLE)
            grid, npix, ids = R.sparse_layout(prm, rect)
            assert grid == (13, 4, 4, 4)
            assert npix == (w * h if r == 0 else -(-16 // (world - 1)) * tile * tile)
            assert (len(ids) == 0) if r == 0 else all((int(i) // 30 - 4) * 4 + int(i) % 30 - 13 >= 0 for i in ids)
            seen += [int(i) for i in ids]
        assert sorted(seen) == want
        # the dense layout: every tile of the frame to ranks 1.., none to rank 0
        total = []
        for r in range(world):
            npix, ids = R.tile_layout(R.make_params(w, h, tile=tile, rank=r, nranks=world, flags=R.ROOT_IDLE))
            assert npix == (0 if r == 0 else -(-510 // (world - 1)) * tile * tile)
            assert r != 0 or len(ids) == 0
            assert (ids % (world - 1) == r - 1).all()
            total += list(ids)
        assert sorted(total) == list(range(510))
    with pytest.raises(R.InvertError):
        R.tile_layout(R.make_params(w, h, rank=1, nranks=2, flags=R.ROOT_IDLE))  # needs a rank beside rank 0 to share with
    # an empty rectangle deals nothing
    grid, npix, ids = R.sparse_layout(R.make_params(w, h, rank=1, nranks=2, flags=R.SPARSE), (0, 0, 0, 0))
    assert grid[2] * grid[3] == 0 and npix == 0 and len(ids) == 0
    with pytest.raises(R.InvertError):
        R.sparse_layout(R.make_params(w, h, rank=0, nranks=2), (10, 10, 5, 20))
    with pytest.raises(R.InvertError):
        R.sparse_layout(R.make_params(w, h), rect)  # one rank: nothing to composite
