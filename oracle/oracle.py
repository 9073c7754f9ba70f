"""ctypes binding of the CPU oracle (oracle/liboracle.so) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")

LAMBERT, INVERT, RGBA, BRUTE_FORCE = 1, 2, 4, 8


class Camera(C.Structure):
    _fields_ = [("model", C.c_float * 16), ("view", C.c_float * 16), ("proj", C.c_float * 16),
                ("cam_angle", C.c_float * 3)]


class Params(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32), ("ao_rays", C.c_uint32),
                ("flags", C.c_uint32), ("seed", C.c_uint32), ("tile", C.c_uint32), ("rank", C.c_uint32),
                ("nranks", C.c_uint32), ("ao_tmax", C.c_float)]


class Counters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("box_tests", C.c_uint64), ("tri_tests", C.c_uint64), ("hits", C.c_uint64)]


def build() -> str:
    """Compile the oracle with gcc if the library is missing or stale."""
    src = [os.path.join(_HERE, f) for f in ("oracle.cpp", "oracle.h", "srgb_thresholds.inc", "ao_dirs.inc")]
    if not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        fp, up, u8p = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)
        L.orc_load_obj.argtypes = [C.c_char_p, C.POINTER(fp), up, C.POINTER(up), up]
        L.orc_parse_obj.argtypes = [C.c_char_p, C.c_uint64, C.POINTER(fp), up, C.POINTER(up), up]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_camera_reference.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(Camera)]
        L.orc_camera_reference.restype = None
        L.orc_camera_turntable.argtypes = [C.c_uint32, C.c_uint32, C.c_float, C.POINTER(Camera)]
        L.orc_camera_turntable.restype = None
        L.orc_vertex_shader.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Camera), C.c_void_p, C.c_void_p]
        L.orc_vertex_shader.restype = None
        L.orc_frame_setup.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Camera), C.c_void_p]
        common = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(Camera), C.POINTER(Params),
                  C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_render_bruteforce.argtypes = common
        L.orc_render_bvh.argtypes = common + [C.POINTER(Counters)]
        L.orc_raster.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(Camera), C.c_uint32,
                                 C.c_uint32, C.c_void_p, C.c_void_p, up]
        L.orc_lbvh_build.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32] + [C.c_void_p] * 6
        L.orc_ao_directions.argtypes = [fp, fp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_ao_directions.restype = None
        L.orc_srgb8.argtypes = [C.c_float]
        L.orc_srgb8.restype = C.c_uint8
        L.orc_hash32.argtypes = [C.c_uint32]
        L.orc_hash32.restype = C.c_uint32
        L.orc_sample_offset.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, fp, fp]
        L.orc_sample_offset.restype = None
        _lib = L
    return _lib


class OracleError(RuntimeError):
    pass


def _chk(rc, what):
    if rc != 0:
        raise OracleError("%s failed with code %d" % (what, rc))


def _take(ptr, count, dtype):
    arr = np.ctypeslib.as_array(ptr, shape=(count,)).astype(dtype, copy=True) if count else np.zeros(0, dtype)
    lib().orc_free(ptr)
    return arr


def load_obj(path: str):
    """A1.  Returns (xyzw float32 [nv,4], idx uint32 [nidx])."""
    fp, up = C.POINTER(C.c_float)(), C.POINTER(C.c_uint32)()
    nv, nidx = C.c_uint32(), C.c_uint32()
    _chk(lib().orc_load_obj(path.encode(), C.byref(fp), C.byref(nv), C.byref(up), C.byref(nidx)), "orc_load_obj")
    return _take(fp, nv.value * 4, np.float32).reshape(-1, 4), _take(up, nidx.value, np.uint32)


def parse_obj(data: bytes):
    fp, up = C.POINTER(C.c_float)(), C.POINTER(C.c_uint32)()
    nv, nidx = C.c_uint32(), C.c_uint32()
    _chk(lib().orc_parse_obj(data, len(data), C.byref(fp), C.byref(nv), C.byref(up), C.byref(nidx)), "orc_parse_obj")
    return _take(fp, nv.value * 4, np.float32).reshape(-1, 4), _take(up, nidx.value, np.uint32)


def camera_reference(w: int, h: int) -> Camera:
    cam = Camera()
    lib().orc_camera_reference(w, h, C.byref(cam))
    return cam


def camera_turntable(w: int, h: int, elapsed_seconds: float) -> Camera:
    cam = Camera()
    lib().orc_camera_turntable(w, h, elapsed_seconds, C.byref(cam))
    return cam


def make_params(width, height, spp=1, ao_rays=0, flags=0, seed=0, tile=0, rank=0, nranks=1, ao_tmax=0.0) -> Params:
    return Params(width, height, spp, ao_rays, flags, seed, tile, rank, nranks, ao_tmax)


def vertex_shader(xyzw: np.ndarray, cam: Camera):
    xyzw = np.ascontiguousarray(xyzw, np.float32)
    nv = xyzw.shape[0]
    clip = np.zeros((nv, 4), np.float32)
    col = np.zeros((nv, 3), np.float32)
    lib().orc_vertex_shader(xyzw.ctypes.data, nv, C.byref(cam), clip.ctypes.data, col.ctypes.data)
    return clip, col


def frame_setup(xyzw: np.ndarray, cam: Camera) -> np.ndarray:
    xyzw = np.ascontiguousarray(xyzw, np.float32)
    out = np.zeros(17, np.float32)
    _chk(lib().orc_frame_setup(xyzw.ctypes.data, xyzw.shape[0], C.byref(cam), out.ctypes.data), "orc_frame_setup")
    return out


def _render(fn, xyzw, idx, cam, prm, rect, nthreads, extra=()):
    xyzw = np.ascontiguousarray(xyzw, np.float32)
    idx = np.ascontiguousarray(idx, np.uint32)
    w, h = prm.width, prm.height
    x0, y0, x1, y1 = rect if rect is not None else (0, 0, w, h)
    # Outside the rectangle the buffers keep the clear values.
    color = np.zeros((h, w, 4), np.uint8)
    color[..., 3] = 255
    if prm.flags & INVERT:
        color[..., :3] = 255
    depth = np.ones((h, w), np.float32)
    primid = np.full((h, w), 0xFFFFFFFF, np.uint32)
    rc = fn(xyzw.ctypes.data, xyzw.shape[0], idx.ctypes.data, idx.size, C.byref(cam), C.byref(prm), x0, y0, x1, y1,
            color.ctypes.data, depth.ctypes.data, primid.ctypes.data, nthreads, *extra)
    _chk(rc, "oracle render")
    return color, depth, primid


def render_bruteforce(xyzw, idx, cam, prm, rect=None, nthreads=None):
    nthreads = nthreads or min(16, os.cpu_count() or 1)
    return _render(lib().orc_render_bruteforce, xyzw, idx, cam, prm, rect, nthreads)


def render_bvh(xyzw, idx, cam, prm, rect=None, nthreads=None, counters: Counters | None = None):
    nthreads = nthreads or min(16, os.cpu_count() or 1)
    cnt = counters if counters is not None else Counters()
    return _render(lib().orc_render_bvh, xyzw, idx, cam, prm, rect, nthreads, (C.byref(cnt),))


def raster(xyzw, idx, cam, w, h):
    xyzw = np.ascontiguousarray(xyzw, np.float32)
    idx = np.ascontiguousarray(idx, np.uint32)
    depth = np.zeros((h, w), np.float64)
    primid = np.zeros((h, w), np.uint32)
    nfront = C.c_uint32()
    _chk(lib().orc_raster(xyzw.ctypes.data, xyzw.shape[0], idx.ctypes.data, idx.size, C.byref(cam), w, h,
                          depth.ctypes.data, primid.ctypes.data, C.byref(nfront)), "orc_raster")
    return depth, primid, nfront.value


def lbvh_build(xyzw, idx):
    xyzw = np.ascontiguousarray(xyzw, np.float32)
    idx = np.ascontiguousarray(idx, np.uint32)
    n = idx.size // 3
    m = max(n - 1, 0)
    out = dict(morton=np.zeros(n, np.uint64), order=np.zeros(n, np.uint32), child=np.zeros((m, 2), np.uint32),
               parent=np.zeros(m, np.uint32), leaf_parent=np.zeros(n, np.uint32), box=np.zeros((m, 12), np.float32))
    _chk(lib().orc_lbvh_build(xyzw.ctypes.data, xyzw.shape[0], idx.ctypes.data, idx.size, out["morton"].ctypes.data,
                              out["order"].ctypes.data, out["child"].ctypes.data, out["parent"].ctypes.data,
                              out["leaf_parent"].ctypes.data, out["box"].ctypes.data), "orc_lbvh_build")
    return out


def hash32(x: int) -> int:
    return int(lib().orc_hash32(x & 0xFFFFFFFF))


def ao_directions(e1, e2, pix: int, seed: int, rays: int) -> np.ndarray:
    """Directions of the occlusion rays of pixel `pix` on a triangle with edges e1, e2: float32 [rays, 3]."""
    a = (C.c_float * 3)(*[float(v) for v in e1])
    b = (C.c_float * 3)(*[float(v) for v in e2])
    out = np.zeros((rays, 3), np.float32)
    lib().orc_ao_directions(a, b, pix, seed, rays, out.ctypes.data)
    return out


def srgb8(x: float) -> int:
    return int(lib().orc_srgb8(float(x)))


def sample_offset(pixel: int, sample: int, seed: int):
    sx, sy = C.c_float(), C.c_float()
    lib().orc_sample_offset(pixel, sample, seed, C.byref(sx), C.byref(sy))
    return sx.value, sy.value
