"""ctypes binding of libinvert_hip.so (include/invert.h).

This is plumbing around the C ABI: device memory comes from the caller (numpy host
arrays or torch device tensors), every pixel is computed by the HIP library.  There
is no CPU fallback: a missing library or a missing GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# INVERT_HIP_LIB names another build of the same library (tuning experiments); the default is the in-tree build
LIB_PATH = os.environ.get("INVERT_HIP_LIB") or os.path.join(_HERE, "libinvert_hip.so")

LAMBERT, INVERT, RGBA, BRUTE_FORCE, LANE_TRAVERSAL, SPARSE, BLOCKS, ROOT_IDLE, TILE_TRAVERSAL = 1, 2, 4, 8, 16, 32, 64, 128, 256
PATH_BRUTE, PATH_LANES, PATH_BLOCKS, PATH_TILES, PATH_TILES_BATCHED, PATH_SCATTER, PATH_BINNED = 0, 1, 2, 3, 4, 5, 16
MAX_FRAMES_IN_FLIGHT = 4  # INV_MAX_FRAMES_IN_FLIGHT
CONTEXT_STREAM = 0xFFFFFFFF  # INV_CONTEXT_STREAM
OK, ERR_ARG, ERR_IO, ERR_FORMAT, ERR_NOMEM, ERR_HIP, ERR_UNSUPPORTED, ERR_NO_DEVICE = range(8)

# every symbol include/invert.h declares
SYMBOLS = (
    "inv_ctx_create", "inv_ctx_destroy", "inv_ctx_synchronize", "inv_ctx_trim", "inv_ctx_stream", "inv_ctx_device",
    "inv_comm_unique_id", "inv_comm_create_rccl", "inv_comm_create_host", "inv_comm_gather", "inv_comm_destroy",
    "inv_mesh_load_obj", "inv_mesh_load_obj_general", "inv_mesh_parse_obj", "inv_mesh_from_arrays", "inv_mesh_counts", "inv_mesh_vertices",
    "inv_mesh_indices", "inv_mesh_pin", "inv_mesh_destroy",
    "inv_camera_reference", "inv_camera_turntable",
    "inv_scene_build", "inv_scene_build_instanced", "inv_scene_counts", "inv_scene_destroy", "inv_scene_read_lbvh", "inv_scene_read_fine_cut",
    "inv_vertex_shader", "inv_render", "inv_render_noalloc", "inv_draw_path", "inv_tile_layout", "inv_deinterleave",
    "inv_render_batch", "inv_render_in_flight", "inv_frame_wait", "inv_frames_join", "inv_frame_stream",
    "inv_scene_hit_rect", "inv_sparse_layout", "inv_deinterleave_sparse",
    "inv_write_ppm", "inv_write_png", "inv_write_pfm", "inv_write_primid",
    "inv_last_error", "inv_version",
)


class Camera(C.Structure):
    """inv_camera: UniformBufferObject + push constant (reference src/invert.cpp:266-271, 633)."""
    _fields_ = [("model", C.c_float * 16), ("view", C.c_float * 16), ("proj", C.c_float * 16),
                ("cam_angle", C.c_float * 3)]


class Params(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32), ("ao_rays", C.c_uint32),
                ("flags", C.c_uint32), ("seed", C.c_uint32), ("tile", C.c_uint32), ("rank", C.c_uint32),
                ("nranks", C.c_uint32), ("ao_tmax", C.c_float)]


class Outputs(C.Structure):
    _fields_ = [("color", C.c_void_p), ("depth", C.c_void_p), ("primid", C.c_void_p), ("on_device", C.c_int)]


class InvertError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__("inv error %d: %s" % (code, message))
        self.code = code


_lib = None


def lib():
    """Load the HIP library; raises if it has not been built (there is no other implementation)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: build it with `make -C renderer_amd/csrc` (hipcc, gfx950); "
                          "there is no CPU fallback" % LIB_PATH)
    # torch bundles its own HIP runtime (same SONAME as /opt/rocm's).  Loading it first makes this library bind
    # to that copy; two HIP runtimes in one process cannot both open the GPU.
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, u32p, u64p = C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    L.inv_last_error.restype = C.c_char_p
    L.inv_version.restype = C.c_char_p
    L.inv_ctx_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
    L.inv_ctx_destroy.argtypes = [vp]
    L.inv_ctx_synchronize.argtypes = [vp]
    L.inv_ctx_trim.argtypes = [vp]
    L.inv_ctx_stream.argtypes = [vp, C.POINTER(vp)]
    L.inv_ctx_device.argtypes = [vp, C.POINTER(C.c_int)]
    L.inv_comm_unique_id.argtypes = [vp]
    L.inv_comm_create_rccl.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.POINTER(vp)]
    L.inv_comm_create_host.argtypes = [vp, C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(vp)]
    L.inv_comm_gather.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64]
    L.inv_comm_destroy.argtypes = [vp]
    L.inv_mesh_load_obj.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.inv_mesh_load_obj_general.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.inv_write_ppm.argtypes = [C.c_char_p, vp, C.c_uint32, C.c_uint32, C.c_uint32]
    L.inv_write_png.argtypes = [C.c_char_p, vp, C.c_uint32, C.c_uint32, C.c_uint32]
    L.inv_write_pfm.argtypes = [C.c_char_p, vp, C.c_uint32, C.c_uint32]
    L.inv_write_primid.argtypes = [C.c_char_p, vp, C.c_uint32, C.c_uint32]
    L.inv_mesh_parse_obj.argtypes = [C.c_char_p, C.c_uint64, C.POINTER(vp)]
    L.inv_mesh_from_arrays.argtypes = [vp, C.c_uint32, vp, C.c_uint32, C.POINTER(vp)]
    L.inv_mesh_counts.argtypes = [vp, u32p, u32p]
    L.inv_mesh_vertices.argtypes = [vp]
    L.inv_mesh_vertices.restype = C.POINTER(C.c_float)
    L.inv_mesh_indices.argtypes = [vp]
    L.inv_mesh_indices.restype = u32p
    L.inv_mesh_pin.argtypes = [vp]
    L.inv_mesh_destroy.argtypes = [vp]
    L.inv_camera_reference.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(Camera)]
    L.inv_camera_turntable.argtypes = [C.c_uint32, C.c_uint32, C.c_float, C.POINTER(Camera)]
    L.inv_scene_build.argtypes = [vp, vp, C.POINTER(vp)]
    L.inv_scene_build_instanced.argtypes = [vp, vp, vp, C.c_uint32, C.POINTER(vp)]
    L.inv_scene_counts.argtypes = [vp, u32p, u32p, u64p]
    L.inv_scene_destroy.argtypes = [vp]
    L.inv_scene_read_lbvh.argtypes = [vp] + [vp] * 6
    L.inv_scene_read_fine_cut.argtypes = [vp, vp, C.c_uint32, u32p]
    L.inv_vertex_shader.argtypes = [vp, C.POINTER(Camera), vp, vp, C.c_int]
    L.inv_render.argtypes = [vp, C.POINTER(Camera), C.POINTER(Params), C.POINTER(Outputs)]
    L.inv_draw_path.argtypes = [vp, C.POINTER(Camera), C.POINTER(Params), u32p]
    L.inv_render_noalloc.argtypes = [vp, C.POINTER(Camera), C.POINTER(Params), C.POINTER(Outputs)]
    L.inv_render_in_flight.argtypes = [vp, C.c_uint32, C.POINTER(Camera), C.POINTER(Params), C.POINTER(Outputs)]
    L.inv_render_batch.argtypes = [vp, C.c_uint32, C.c_uint32, vp, vp, vp]
    L.inv_frame_wait.argtypes = [vp, C.c_uint32]
    L.inv_frames_join.argtypes = [vp]
    L.inv_frame_stream.argtypes = [vp, C.c_uint32, C.POINTER(vp)]
    L.inv_tile_layout.argtypes = [C.POINTER(Params), u64p, u32p, C.c_uint32, u32p]
    L.inv_deinterleave.argtypes = [vp, C.POINTER(Params), vp, C.c_uint64, vp, C.c_uint32]
    rect = C.POINTER(C.c_uint32 * 4)
    L.inv_scene_hit_rect.argtypes = [vp, C.POINTER(Camera), C.c_uint32, C.c_uint32, rect]
    L.inv_sparse_layout.argtypes = [C.POINTER(Params), rect, rect, u64p, u32p, C.c_uint32, u32p]
    L.inv_deinterleave_sparse.argtypes = [vp, C.POINTER(Params), rect, vp, C.c_uint64, C.c_uint64, C.c_uint32, vp,
                                          C.c_uint64]
    _lib = L
    return L


def _chk(rc: int) -> None:
    if rc != 0:
        raise InvertError(rc, lib().inv_last_error().decode(errors="replace"))


def version() -> str:
    return lib().inv_version().decode()


def make_params(width, height, spp=1, ao_rays=0, flags=0, seed=0, tile=0, rank=0, nranks=1, ao_tmax=0.0) -> Params:
    return Params(width, height, spp, ao_rays, flags, seed, tile, rank, nranks, ao_tmax)


def camera_reference(width: int, height: int) -> Camera:
    """The reference's hard-coded camera (src/invert.cpp:630-639)."""
    cam = Camera()
    _chk(lib().inv_camera_reference(width, height, C.byref(cam)))
    return cam


def camera_turntable(width: int, height: int, elapsed_seconds: float) -> Camera:
    """The reference camera with the model turned about the up axis by elapsed_seconds * 90 degrees."""
    cam = Camera()
    _chk(lib().inv_camera_turntable(width, height, elapsed_seconds, C.byref(cam)))
    return cam


class Mesh:
    """Host mesh: the reference's struct Obj (src/invert.cpp:25-39)."""

    def __init__(self, handle):
        self._h = handle

    @staticmethod
    def load_obj(path: str) -> "Mesh":
        h = C.c_void_p()
        _chk(lib().inv_mesh_load_obj(os.fsencode(path), C.byref(h)))
        return Mesh(h)

    @staticmethod
    def load_obj_general(path: str) -> "Mesh":
        """Tolerant reader (free-form numbers, i/t/n corners, polygons, CRLF); not bit-identical to load_obj."""
        h = C.c_void_p()
        _chk(lib().inv_mesh_load_obj_general(os.fsencode(path), C.byref(h)))
        return Mesh(h)

    @staticmethod
    def parse_obj(text: bytes) -> "Mesh":
        h = C.c_void_p()
        _chk(lib().inv_mesh_parse_obj(text, len(text), C.byref(h)))
        return Mesh(h)

    @staticmethod
    def from_arrays(xyzw: np.ndarray, idx: np.ndarray) -> "Mesh":
        xyzw = np.ascontiguousarray(xyzw, np.float32).reshape(-1, 4)
        idx = np.ascontiguousarray(idx, np.uint32).reshape(-1)
        h = C.c_void_p()
        _chk(lib().inv_mesh_from_arrays(xyzw.ctypes.data, xyzw.shape[0], idx.ctypes.data, idx.size, C.byref(h)))
        return Mesh(h)

    def counts(self):
        nv, nidx = C.c_uint32(), C.c_uint32()
        _chk(lib().inv_mesh_counts(self._h, C.byref(nv), C.byref(nidx)))
        return nv.value, nidx.value

    def pin(self) -> "Mesh":
        """Move the arrays into page-locked host memory: scene builds then upload by DMA straight out of them."""
        _chk(lib().inv_mesh_pin(self._h))
        return self

    @property
    def vertices(self) -> np.ndarray:
        nv, _ = self.counts()
        if nv == 0:
            return np.zeros((0, 4), np.float32)
        return np.ctypeslib.as_array(lib().inv_mesh_vertices(self._h), shape=(nv * 4,)).reshape(nv, 4).copy()

    @property
    def indices(self) -> np.ndarray:
        _, nidx = self.counts()
        if nidx == 0:
            return np.zeros(0, np.uint32)
        return np.ctypeslib.as_array(lib().inv_mesh_indices(self._h), shape=(nidx,)).copy()

    def close(self):
        if self._h:
            lib().inv_mesh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_ppm(path: str, color: np.ndarray, flags: int = 0) -> None:
    color = np.ascontiguousarray(color, np.uint8)
    h, w = color.shape[:2]
    _chk(lib().inv_write_ppm(os.fsencode(path), color.ctypes.data, w, h, flags))


def write_png(path: str, color: np.ndarray, flags: int = 0) -> None:
    color = np.ascontiguousarray(color, np.uint8)
    h, w = color.shape[:2]
    _chk(lib().inv_write_png(os.fsencode(path), color.ctypes.data, w, h, flags))


def write_pfm(path: str, depth: np.ndarray) -> None:
    depth = np.ascontiguousarray(depth, np.float32)
    h, w = depth.shape
    _chk(lib().inv_write_pfm(os.fsencode(path), depth.ctypes.data, w, h))


def write_primid(path: str, primid: np.ndarray) -> None:
    primid = np.ascontiguousarray(primid, np.uint32)
    h, w = primid.shape
    _chk(lib().inv_write_primid(os.fsencode(path), primid.ctypes.data, w, h))


class Context:
    """One GPU + one stream (pass torch's current stream handle to share it)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        h = C.c_void_p()
        _chk(lib().inv_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(h)))
        self._h = h
        self.device = device
        s = C.c_void_p()
        _chk(lib().inv_ctx_stream(h, C.byref(s)))
        self.stream = s.value or 0  # hipStream_t the context enqueues on (the one passed in, or its own)

    def synchronize(self):
        _chk(lib().inv_ctx_synchronize(self._h))

    def trim(self):
        """Release the device memory kept from the last scene build for the next one."""
        _chk(lib().inv_ctx_trim(self._h))

    def frame_wait(self, slot: int):
        """Block until every draw enqueued on frame slot `slot` has finished (the reference's fence wait)."""
        _chk(lib().inv_frame_wait(self._h, slot))

    def frames_join(self):
        """Order the context's stream after everything enqueued on the frame slots so far (no host wait)."""
        _chk(lib().inv_frames_join(self._h))

    def frame_stream(self, slot: int) -> int:
        """hipStream_t handle of frame slot `slot` (created on first use)."""
        s = C.c_void_p()
        _chk(lib().inv_frame_stream(self._h, slot, C.byref(s)))
        return s.value or 0

    def deinterleave(self, prm: Params, gathered_ptr: int, frame_ptr: int, rank_stride: int = 0):
        _chk(lib().inv_deinterleave(self._h, C.byref(prm), C.c_void_p(gathered_ptr), rank_stride,
                                    C.c_void_p(frame_ptr), 4))

    def deinterleave_sparse(self, prm: Params, rect, gathered_ptr: int, rank_stride: int, plane_stride: int,
                            nplanes: int, frame_ptr: int, frame_plane_stride: int):
        """Rank 0 of a sparse composite: copy the other ranks' tiles into the row-major frame (all planes at once)."""
        r = (C.c_uint32 * 4)(*rect)
        _chk(lib().inv_deinterleave_sparse(self._h, C.byref(prm), C.byref(r), C.c_void_p(gathered_ptr or None),
                                           rank_stride, plane_stride, nplanes, C.c_void_p(frame_ptr),
                                           frame_plane_stride))

    def close(self):
        """inv_ctx_destroy; raises while scenes built on this context are still open (close them first)."""
        if self._h:
            _chk(lib().inv_ctx_destroy(self._h))
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


COMM_ID_BYTES = 128  # INV_COMM_ID_BYTES


def comm_unique_id() -> bytes:
    """Rank 0: the RCCL id every rank's Comm.rccl needs (hand it to the other ranks, e.g. with a broadcast)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _chk(lib().inv_comm_unique_id(buf))
    return buf.raw


class Comm:
    """The exchange step of a multi-GPU composite through the C ABI (inv_comm_*): RCCL, or host staging files."""

    def __init__(self, handle, ctx, rank, nranks):
        self._h, self.ctx, self.rank, self.nranks = handle, ctx, rank, nranks

    @staticmethod
    def rccl(ctx: "Context", unique_id: bytes, rank: int, nranks: int) -> "Comm":
        h = C.c_void_p()
        _chk(lib().inv_comm_create_rccl(ctx._h, C.c_char_p(unique_id), rank, nranks, C.byref(h)))
        return Comm(h, ctx, rank, nranks)

    @staticmethod
    def host(ctx: "Context", directory: str, rank: int, nranks: int) -> "Comm":
        h = C.c_void_p()
        _chk(lib().inv_comm_create_host(ctx._h, os.fsencode(directory), rank, nranks, C.byref(h)))
        return Comm(h, ctx, rank, nranks)

    def gather(self, send_ptr: int, count: int, recv_ptr: int = 0, rank_stride: int = 0):
        """Every rank > 0 sends `count` 4-byte elements; rank 0 receives rank r's at recv + r * rank_stride."""
        _chk(lib().inv_comm_gather(self._h, C.c_void_p(send_ptr or None), count, C.c_void_p(recv_ptr or None), rank_stride))

    def close(self):
        if self._h:
            lib().inv_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def tile_layout(prm: Params):
    """(pixels stored by this rank, tile ids in storage order)."""
    npix, nt = C.c_uint64(), C.c_uint32()
    _chk(lib().inv_tile_layout(C.byref(prm), C.byref(npix), None, 0, C.byref(nt)))
    ids = np.zeros(max(nt.value, 1), np.uint32)
    _chk(lib().inv_tile_layout(C.byref(prm), C.byref(npix), ids.ctypes.data_as(C.POINTER(C.c_uint32)), ids.size,
                               C.byref(nt)))
    return npix.value, ids[:nt.value]


def sparse_layout(prm: Params, rect):
    """Sparse composite layout for prm.rank: (grid (gx0, gy0, tiles_x, tiles_y), elements per plane, frame tile ids)."""
    r, grid = (C.c_uint32 * 4)(*rect), (C.c_uint32 * 4)()
    npix, nt = C.c_uint64(), C.c_uint32()
    _chk(lib().inv_sparse_layout(C.byref(prm), C.byref(r), C.byref(grid), C.byref(npix), None, 0, C.byref(nt)))
    ids = np.zeros(max(nt.value, 1), np.uint32)
    _chk(lib().inv_sparse_layout(C.byref(prm), C.byref(r), C.byref(grid), C.byref(npix),
                                 ids.ctypes.data_as(C.POINTER(C.c_uint32)), ids.size, C.byref(nt)))
    return tuple(grid), npix.value, ids[:nt.value]


class Scene:
    """HBM-resident triangles + LBVH for one mesh (optionally instanced)."""

    def __init__(self, ctx: Context, mesh: Mesh, translations: np.ndarray | None = None):
        h = C.c_void_p()
        if translations is None:
            _chk(lib().inv_scene_build(ctx._h, mesh._h, C.byref(h)))
        else:
            tr = np.ascontiguousarray(translations, np.float32).reshape(-1, 3)
            _chk(lib().inv_scene_build_instanced(ctx._h, mesh._h, tr.ctypes.data, tr.shape[0], C.byref(h)))
        self._h = h
        self.ctx = ctx
        self.nv, _ = mesh.counts()

    def counts(self):
        ntri, nnodes, nbytes = C.c_uint32(), C.c_uint32(), C.c_uint64()
        _chk(lib().inv_scene_counts(self._h, C.byref(ntri), C.byref(nnodes), C.byref(nbytes)))
        return ntri.value, nnodes.value, nbytes.value

    def read_lbvh(self):
        n, _, _ = self.counts()
        m = max(n - 1, 0)
        out = dict(morton=np.zeros(n, np.uint64), order=np.zeros(n, np.uint32), child=np.zeros((m, 2), np.uint32),
                   parent=np.zeros(m, np.uint32), leaf_parent=np.zeros(n, np.uint32), box=np.zeros((m, 12), np.float32))
        _chk(lib().inv_scene_read_lbvh(self._h, out["morton"].ctypes.data, out["order"].ctypes.data,
                                       out["child"].ctypes.data, out["parent"].ctypes.data,
                                       out["leaf_parent"].ctypes.data, out["box"].ctypes.data))
        return out

    def read_fine_cut(self):
        """The fine cut as (box lo [n,3], box hi [n,3], first sorted leaf [n], leaf count [n]); n = 0: the scene has none."""
        n = C.c_uint32()
        _chk(lib().inv_scene_read_fine_cut(self._h, None, 0, C.byref(n)))
        rec = np.zeros((max(n.value, 1), 8), np.float32)
        _chk(lib().inv_scene_read_fine_cut(self._h, rec.ctypes.data, n.value, C.byref(n)))
        rec = rec[:n.value]
        return rec[:, 0:3].copy(), rec[:, 4:7].copy(), rec[:, 3].copy().view(np.uint32), rec[:, 7].copy().view(np.uint32)

    def hit_rect(self, cam: Camera, width: int, height: int):
        """Pixel rectangle (x0, y0, x1, y1) outside which every pixel is background for this camera."""
        r = (C.c_uint32 * 4)()
        _chk(lib().inv_scene_hit_rect(self._h, C.byref(cam), width, height, C.byref(r)))
        return tuple(r)

    def vertex_shader(self, cam: Camera):
        """vertexShader over the mesh (reference src/shader.slang:22-29): (clip [nv,4], colour [nv,3])."""
        clip = np.zeros((self.nv, 4), np.float32)
        col = np.zeros((self.nv, 3), np.float32)
        _chk(lib().inv_vertex_shader(self._h, C.byref(cam), clip.ctypes.data, col.ctypes.data, 0))
        return clip, col

    def render(self, cam: Camera, prm: Params):
        """The draw into host arrays: (colour u8 [h,w,4], depth f32 [h,w], primid u32 [h,w]).

        With nranks > 1 the arrays are this rank's tile slots, shape [slots, tile, tile(,4)].
        """
        sparse = (prm.flags & SPARSE) and (prm.nranks or 1) > 1
        if sparse:
            _, npix, _ = sparse_layout(prm, self.hit_rect(cam, prm.width, prm.height))
        else:
            npix, _ = tile_layout(prm)
        color = np.empty(npix * 4, np.uint8)
        depth = np.empty(npix, np.float32)
        primid = np.empty(npix, np.uint32)
        out = Outputs(color.ctypes.data, depth.ctypes.data, primid.ctypes.data, 0)
        _chk(lib().inv_render(self._h, C.byref(cam), C.byref(prm), C.byref(out)))
        if (prm.nranks or 1) <= 1 or (sparse and prm.rank == 0):
            h, w = prm.height, prm.width
            return color.reshape(h, w, 4), depth.reshape(h, w), primid.reshape(h, w)
        t = prm.tile or 64
        return color.reshape(-1, t, t, 4), depth.reshape(-1, t, t), primid.reshape(-1, t, t)

    def render_device(self, cam: Camera, prm: Params, color_ptr: int = 0, depth_ptr: int = 0, primid_ptr: int = 0):
        """Enqueue the draw into device buffers (e.g. torch tensors' data_ptr()) on the context's stream."""
        key = (color_ptr, depth_ptr, primid_ptr)
        if getattr(self, "_dev_out_key", None) != key:  # reuse the struct between frames into the same buffers
            self._dev_out = Outputs(color_ptr or None, depth_ptr or None, primid_ptr or None, 1)
            self._dev_out_key = key
        rc = lib().inv_render(self._h, C.byref(cam), C.byref(prm), C.byref(self._dev_out))
        if rc:
            _chk(rc)

    def draw_path(self, cam: Camera, prm: Params) -> int:
        """Which kernels inv_render would draw this frame with: a PATH_* value, plus PATH_BINNED."""
        v = C.c_uint32()
        _chk(lib().inv_draw_path(self._h, C.byref(cam), C.byref(prm), C.byref(v)))
        return v.value

    def render_noalloc(self, cam: Camera, prm: Params, color_ptr: int = 0, depth_ptr: int = 0, primid_ptr: int = 0):
        """inv_render_noalloc: the draw into device buffers without any device allocation (static occlusion queue)."""
        out = Outputs(color_ptr or None, depth_ptr or None, primid_ptr or None, 1)
        _chk(lib().inv_render_noalloc(self._h, C.byref(cam), C.byref(prm), C.byref(out)))

    def render_in_flight(self, slot: int, cam: Camera, prm: Params, color_ptr: int = 0, depth_ptr: int = 0,
                         primid_ptr: int = 0):
        """Enqueue the draw into device buffers on frame slot `slot` (a stream of its own): draws on different slots
        overlap on the GPU.  Context.frame_wait / frames_join order the consumers."""
        cache = self.__dict__.setdefault("_flight_out", {})
        key = (slot, color_ptr, depth_ptr, primid_ptr)
        out = cache.get(key)
        if out is None:
            if len(cache) >= 64:  # callers cycle through a few attachment sets; anything else just rebuilds the struct
                cache.clear()
            out = cache[key] = Outputs(color_ptr or None, depth_ptr or None, primid_ptr or None, 1)
        rc = lib().inv_render_in_flight(self._h, slot, C.byref(cam), C.byref(prm), C.byref(out))
        if rc:
            _chk(rc)

    def make_batch(self, cams, prms, ptrs):
        """Argument arrays for render_batch, built once and reused: cams and prms are sequences of Camera / Params,
        ptrs a sequence of (colour, depth, primid) device pointers."""
        n = len(cams)
        return (n, (Camera * n)(*cams), (Params * n)(*prms),
                (Outputs * n)(*[Outputs(c or None, d or None, p or None, 1) for c, d, p in ptrs]))

    def render_batch(self, batch, slot: int = CONTEXT_STREAM):
        """inv_render_batch: enqueue the draws of `batch` (from make_batch) on frame slot `slot` or the context's stream."""
        n, cams, prms, outs = batch
        rc = lib().inv_render_batch(self._h, slot, n, cams, prms, outs)
        if rc:
            _chk(rc)

    def render_in_flight_host(self, slot: int, cam: Camera, prm: Params):
        """inv_render_in_flight into host arrays (the call waits for that slot): (colour, depth, primid) of a whole frame."""
        h, w = prm.height, prm.width
        color, depth, primid = np.empty((h, w, 4), np.uint8), np.empty((h, w), np.float32), np.empty((h, w), np.uint32)
        out = Outputs(color.ctypes.data, depth.ctypes.data, primid.ctypes.data, 0)
        _chk(lib().inv_render_in_flight(self._h, slot, C.byref(cam), C.byref(prm), C.byref(out)))
        return color, depth, primid

    def close(self):
        if self._h:
            lib().inv_scene_destroy(self._h)
            self._h = None

    def __del__(self):  # a scene keeps its context alive (self.ctx), so the context is always closed after it
        try:
            self.close()
        except Exception:
            pass
