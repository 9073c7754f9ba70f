"""Multi-GPU plumbing: one process per GPU, screen tiles dealt round-robin, attachments gathered to rank 0.

Pixels are independent and the scene is replicated (every rank builds the same deterministic LBVH), so the only
exchange is the final composite: each rank's tile-major planes go to rank 0 in one gather, and rank 0 scatters them
into the row-major frame (inv_deinterleave on the GPU).  Works with the `nccl` (= RCCL) backend on device tensors
and with `gloo` on host tensors (used by the CPU tests).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import api


def shard_params(width, height, rank, world, tile=64, **kw) -> api.Params:
    """Params for this rank: it renders the tiles with tile_id % world == rank."""
    return api.make_params(width, height, tile=tile, rank=rank, nranks=world, **kw)


def alloc_planes(prm: api.Params, device) -> torch.Tensor:
    """[3, npix] int32: colour, depth bits and primitive id of this rank's tile slots (4 bytes per pixel each)."""
    npix, _ = api.tile_layout(prm)
    return torch.empty(3, npix, dtype=torch.int32, device=device)


def gather_to_rank0(planes: torch.Tensor, rank: int, world: int, gathered: torch.Tensor | None = None):
    """One collective: every rank's [3, npix] planes -> [world, 3, npix] on rank 0 (None elsewhere)."""
    if world == 1:
        return planes.unsqueeze(0)
    if rank == 0:
        if gathered is None:
            gathered = torch.empty((world,) + tuple(planes.shape), dtype=planes.dtype, device=planes.device)
        dist.gather(planes, list(gathered.unbind(0)), dst=0)
        return gathered
    dist.gather(planes, None, dst=0)
    return None


def deinterleave_on_gpu(ctx: api.Context, prm0: api.Params, gathered: torch.Tensor, frame: torch.Tensor) -> None:
    """Rank 0: [world, 3, npix] tile slots -> [3, h*w] row-major planes, enqueued on the context's stream."""
    npix = gathered.shape[2]
    for k in range(3):  # plane k of rank r starts at gathered[r, k]: ranks are 3 * npix elements apart
        ctx.deinterleave(prm0, gathered[0, k].data_ptr(), frame[k].data_ptr(), rank_stride=3 * npix)
