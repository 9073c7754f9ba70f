"""Multi-GPU plumbing: one process per GPU, screen tiles dealt round-robin, attachments gathered to rank 0.

Pixels are independent and the scene is replicated (every rank builds the same deterministic LBVH), so the only
exchange is the final composite: each rank's tile-major planes go to rank 0 in one gather, and rank 0 scatters them
into the row-major frame on the GPU.  Two layouts:

  dense   every tile of the frame is dealt out (tile_id % world); rank 0 receives the whole frame.
  sparse  only the tiles the scene's projected bounds touch are dealt out; rank 0 keeps the row-major frame,
          clears the rest itself and receives just the other ranks' hit tiles (a few hundred KB instead of the
          frame: xGMI into one GPU is an order of magnitude slower than that GPU's HBM).

Works with the `nccl` (= RCCL) backend on device tensors and with `gloo` on host tensors (used by the CPU tests).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import api


def order_current_stream_after(ctx: api.Context) -> None:
    """Make torch's current stream wait for everything enqueued on the context's stream so far.

    The library draws on the context's stream; torch.distributed enqueues a collective on torch's current stream.
    They are the same stream only if the context was created on it (Context(device, stream=...)); otherwise the
    collective would read tiles that may not have been drawn yet.  An event recorded on the context's stream and
    waited for by the current stream orders them on the device (no host wait; nothing to do when they coincide)."""
    cur = torch.cuda.current_stream()
    if cur.cuda_stream == ctx.stream:
        return
    ev = torch.cuda.Event()
    ev.record(torch.cuda.ExternalStream(ctx.stream))
    cur.wait_event(ev)


def order_context_stream_after_current(ctx: api.Context) -> None:
    """The other direction: what the context enqueues next (the copy of received tiles) waits for the collective."""
    cur = torch.cuda.current_stream()
    if cur.cuda_stream == ctx.stream:
        return
    ev = torch.cuda.Event()
    ev.record(cur)
    torch.cuda.ExternalStream(ctx.stream).wait_event(ev)


def make_comm(ctx: api.Context, rank: int, world: int, transport: str = "rccl") -> api.Comm:
    """The library's own communicator (inv_comm_*: grouped RCCL send / receive on the context's stream, or staging
    files) for the ranks of the initialised torch.distributed group, which only carries the rendezvous here: rank 0's
    RCCL id, or the name of a fresh directory under /dev/shm."""
    if transport == "rccl":
        box = [api.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return api.Comm.rccl(ctx, box[0], rank, world)
    import tempfile
    box = [tempfile.mkdtemp(prefix="inv_comm_", dir="/dev/shm") if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return api.Comm.host(ctx, box[0], rank, world)


def shard_params(width, height, rank, world, tile=64, **kw) -> api.Params:
    """Params for this rank: it renders the tiles with tile_id % world == rank."""
    return api.make_params(width, height, tile=tile, rank=rank, nranks=world, **kw)


def alloc_planes(prm: api.Params, device) -> torch.Tensor:
    """[3, npix] int32: colour, depth bits and primitive id of this rank's tile slots (4 bytes per pixel each)."""
    npix, _ = api.tile_layout(prm)
    if npix == 0 and (prm.flags & api.ROOT_IDLE):  # an idle rank 0 draws nothing, but a gather wants equal shares
        other = api.make_params(prm.width, prm.height, tile=prm.tile, rank=1, nranks=prm.nranks, flags=prm.flags)
        npix, _ = api.tile_layout(other)
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


class SparseComposite:
    """Buffers and steps of the sparse composite for one rank, one camera and one frame size.

        sc = SparseComposite(scene, cam, width, height, rank, world, flags=..., spp=...)
        sc.render()            # every rank: enqueue the draw of its share
        sc.composite()         # every rank: the gather; rank 0 also copies the received tiles into sc.frame
        sc.frame               # rank 0: [3, h*w] int32 planes (colour, depth bits, primitive id); None elsewhere

    The draw and the copy of received tiles run on the context's stream, the collective on torch's current stream.
    With a context created on torch's current stream they are one stream; otherwise gather() orders the two streams
    with events on the device (see order_current_stream_after).  After composite(), sc.frame is complete once the
    context's stream has been waited for (ctx.synchronize(), or an event on it).
    """

    def __init__(self, scene: api.Scene, cam: api.Camera, width: int, height: int, rank: int, world: int,
                 tile: int = 64, device="cuda", comm: api.Comm | None = None, **params):
        if world < 2:
            raise ValueError("the sparse composite needs at least two ranks")
        self.scene, self.cam, self.rank, self.world = scene, cam, rank, world
        self.comm = comm  # None: torch.distributed.gather; else the library's own exchange on the context's stream
        flags = params.pop("flags", 0) | api.SPARSE
        self.prm = api.make_params(width, height, tile=tile, rank=rank, nranks=world, flags=flags, **params)
        self.prm0 = api.make_params(width, height, tile=tile, rank=0, nranks=world, flags=flags)
        self.rect = scene.hit_rect(cam, width, height)
        # every rank > 0 holds the same number of slots; rank 0 sends a dummy of that size (a gather wants equal parts)
        other = api.make_params(width, height, tile=tile, rank=1, nranks=world, flags=flags)
        self.grid, self.slot_elems, _ = api.sparse_layout(other, self.rect)
        self.frame = self.gathered = None
        if rank == 0:
            self.frame = torch.empty(3, width * height, dtype=torch.int32, device=device)
            self.gathered = torch.empty(world, 3, self.slot_elems, dtype=torch.int32, device=device)
            # rank 0's own part of the gather is never read; it only has to exist and not alias the output
            self.planes = torch.zeros(3, self.slot_elems, dtype=torch.int32, device=device)
            self._out = self.frame
        else:
            self.planes = torch.empty(3, self.slot_elems, dtype=torch.int32, device=device)
            self._out = self.planes
        # per-frame calls reuse these instead of rebuilding views and pointers
        self._ptrs = tuple(self._out[k].data_ptr() for k in range(3))
        self._parts = list(self.gathered.unbind(0)) if rank == 0 else None

    def render(self):
        self.scene.render_device(self.cam, self.prm, *self._ptrs)

    def gather(self):
        if self.slot_elems == 0:
            return  # nothing on screen: rank 0's frame is already complete
        if self.comm is not None:  # same stream as the draw and the copy: nothing to order
            n = 3 * self.slot_elems
            self.comm.gather(self.planes.data_ptr(), n, self.gathered.data_ptr() if self.rank == 0 else 0, n)
            return
        if self.planes.is_cuda:
            order_current_stream_after(self.scene.ctx)  # the collective reads what the draw wrote
        if self.rank == 0:
            dist.gather(self.planes, self._parts, dst=0)
        else:
            dist.gather(self.planes, None, dst=0)
        if self.planes.is_cuda:
            order_context_stream_after_current(self.scene.ctx)  # the copy (and the next draw) follow the collective

    def scatter_received(self):
        if self.rank == 0 and self.slot_elems:
            n = self.slot_elems
            self.scene.ctx.deinterleave_sparse(self.prm0, self.rect, self.gathered.data_ptr(), 3 * n, n, 3,
                                               self.frame.data_ptr(), self.frame.shape[1])

    def composite(self):
        self.gather()
        self.scatter_received()


class SparseCompositeBatch:
    """The sparse composite for a stream of frames: `frames` frames per submission and per collective, and `sets`
    sets of buffers, so that the gather of one group of frames overlaps the draws of the next.

        cb = SparseCompositeBatch(scene, cam, width, height, rank, world, frames=8, flags=...)
        for g in range(groups):
            k = g % cb.nsets
            cb.render(k)        # every rank: the draws of a group, one submission on frame slot k
            cb.composite(k)     # every rank: one gather; rank 0 also copies the received tiles into cb.frames[k]
        cb.frames[k]            # rank 0: [frames, 3, h*w] int32 planes (colour, depth bits, primitive id)

    The draws of set k run on frame slot k of the scene's context, the copy of received tiles on the context's stream
    and the gather on torch's current stream (ordered against the context's stream with events on the device when
    they differ).  A set's buffers are drawn into again only after its composite has finished (the slot's stream
    waits for an event recorded behind the composite).
    """

    def __init__(self, scene: api.Scene, cam: api.Camera, width: int, height: int, rank: int, world: int,
                 frames: int = 8, sets: int = 2, tile: int = 32, device="cuda", comm: api.Comm | None = None, **params):
        if world < 2:
            raise ValueError("the sparse composite needs at least two ranks")
        self.comm = comm  # None: torch.distributed.gather; else the library's own exchange on the context's stream
        if not 1 <= sets <= api.MAX_FRAMES_IN_FLIGHT:
            raise ValueError("sets must be between 1 and %d" % api.MAX_FRAMES_IN_FLIGHT)
        self.scene, self.cam, self.rank, self.world, self.nframes, self.nsets = scene, cam, rank, world, frames, sets
        self.width, self.height = width, height
        flags = params.pop("flags", 0) | api.SPARSE
        self.prm = api.make_params(width, height, tile=tile, rank=rank, nranks=world, flags=flags, **params)
        self.prm0 = api.make_params(width, height, tile=tile, rank=0, nranks=world, flags=flags)
        self.rect = scene.hit_rect(cam, width, height)
        other = api.make_params(width, height, tile=tile, rank=1, nranks=world, flags=flags)
        self.grid, self.slot_elems, _ = api.sparse_layout(other, self.rect)
        n, F = self.slot_elems, frames
        self.frames, self.gathered, self.planes, self._batches, self._parts = [], [], [], [], []
        for k in range(sets):
            if rank == 0:
                self.frames.append(torch.empty(F, 3, width * height, dtype=torch.int32, device=device))
                self.gathered.append(torch.empty(world, F, 3, n, dtype=torch.int32, device=device))
                # rank 0's own part of the gather is never read; it only has to exist
                self.planes.append(torch.zeros(F, 3, n, dtype=torch.int32, device=device))
                outs = self.frames[k]
            else:
                self.frames.append(None)
                self.gathered.append(None)
                self.planes.append(torch.empty(F, 3, n, dtype=torch.int32, device=device))
                outs = self.planes[k]
            ptrs = [tuple(outs[j, p].data_ptr() for p in range(3)) for j in range(F)]
            # batches of 1 .. F frames (the last group of a run may be short)
            self._batches.append([scene.make_batch([cam] * m, [self.prm] * m, ptrs[:m]) for m in range(1, F + 1)])
        self.slot_streams = [torch.cuda.ExternalStream(scene.ctx.frame_stream(k)) for k in range(sets)]
        self._done = [torch.cuda.Event() for _ in range(sets)]
        torch.cuda.synchronize()  # the fills above ran on torch's stream, which the slots do not wait for

    def render(self, k: int, n: int | None = None):
        """Enqueue the draws of the first n (default: all) frames of set k on frame slot k."""
        n = self.nframes if n is None else n
        self.scene.render_batch(self._batches[k][n - 1], slot=k)

    def gather(self, k: int, n: int | None = None):
        """The collective of set k on torch's current stream, behind the set's draws."""
        n = self.nframes if n is None else n
        self.scene.ctx.frames_join()  # the context's stream waits for the slots' draws
        if self.slot_elems == 0:
            return  # nothing on screen: rank 0's frames are already complete
        if self.comm is not None:  # the first n frames of every rank's part: n * 3 * s elements, parts F * 3 * s apart
            s, F = self.slot_elems, self.nframes
            self.comm.gather(self.planes[k].data_ptr(), n * 3 * s, self.gathered[k].data_ptr() if self.rank == 0 else 0, F * 3 * s)
            return
        order_current_stream_after(self.scene.ctx)
        if self.rank == 0:
            dist.gather(self.planes[k][:n], [self.gathered[k][r, :n] for r in range(self.world)], dst=0)
        else:
            dist.gather(self.planes[k][:n], None, dst=0)
        order_context_stream_after_current(self.scene.ctx)

    def scatter_received(self, k: int, n: int | None = None):
        """Rank 0: copy the other ranks' tiles of set k into its frames (all planes of all frames in one launch);
        every rank: let the set's slot wait for the composite before it is drawn into again."""
        n = self.nframes if n is None else n
        if self.rank == 0 and self.slot_elems:
            s, F = self.slot_elems, self.nframes
            self.scene.ctx.deinterleave_sparse(self.prm0, self.rect, self.gathered[k].data_ptr(), F * 3 * s, s, 3 * n,
                                               self.frames[k].data_ptr(), self.width * self.height)
        self._done[k].record(torch.cuda.ExternalStream(self.scene.ctx.stream))  # behind the copy on the context's stream
        self.slot_streams[k].wait_event(self._done[k])

    def composite(self, k: int, n: int | None = None):
        self.gather(k, n)
        self.scatter_received(k, n)
