"""CPU tests of the product's host side: loader, camera, argument checks, exported symbols.

No compute entry point is called here (there is no GPU); the parity tests proper are the gpu-marked ones.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

import common
import objfmt
import renderer_amd as R
from renderer_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "invert.h")).read()
    declared = set(re.findall(r"\b(inv_[a-z0-9_]+)\s*\(", header))
    assert declared == set(api.SYMBOLS), declared ^ set(api.SYMBOLS)
    lib = C.CDLL(api.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert "gfx950" in R.version()


def test_struct_layouts_match_the_header():
    assert C.sizeof(api.Camera) == 4 * (16 * 3 + 3)
    assert C.sizeof(api.Params) == 40
    assert C.sizeof(api.Outputs) == 32


def test_loader_is_bit_identical_to_the_oracle(oracle, bunny_path, bunny_mesh):
    m = R.Mesh.load_obj(bunny_path)
    assert m.counts() == (2503, 14904)
    xyzw, idx = bunny_mesh
    assert np.array_equal(m.vertices[:, :3].view(np.uint32), xyzw[:, :3].view(np.uint32))
    assert np.array_equal(m.indices, idx)
    m.close()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_loader_on_synthetic_text(oracle, seed):
    """Random coordinates over the whole exponent range the format allows, negative zero, CRLF face lines."""
    rng = np.random.default_rng(seed)
    nv = 300
    fo = objfmt.FixedObj(
        neg=rng.random((nv, 3)) < 0.5,
        mant=rng.integers(0, 100_000_000, (nv, 3)).astype(np.int32),
        eneg=rng.random((nv, 3)) < 0.7,
        expo=rng.integers(0, 40, (nv, 3)).astype(np.int16),
        faces=rng.integers(1, nv + 1, (200, 3)).astype(np.uint32),
        header=b"# synthetic\n\n# blank line above\n")
    fo.mant[0] = 0  # +-0.0000000
    text = fo.text()
    if seed == 1:  # the Stanford original has CRLF on face lines (src/invert.cpp:178-180)
        text = re.sub(rb"(f [0-9 ]+)\n", rb"\1\r\n", text)
    want_v, want_i = oracle.parse_obj(text)
    m = R.Mesh.parse_obj(text)
    assert np.array_equal(m.vertices[:, :3].view(np.uint32), want_v[:, :3].view(np.uint32))
    assert np.array_equal(m.indices, want_i)
    assert np.isfinite(m.vertices).all() or seed >= 0


@pytest.mark.parametrize("text", [
    b"",                                                   # empty
    b"v 1.0000000e000 1.0000000e000 1.0000000e000",        # no final LF
    b"v 1.0 2.0 3.0\n",                                    # free-form floats
    b"v 1.0000000e+000 1.0000000e000 1.0000000e000\n",     # '+' exponent (src/invert.cpp:120-125 does not handle it)
    b"vn 1.0000000e000 1.0000000e000 1.0000000e000\n",     # first-byte 'v' that is not a vertex line
    b"v 1.0000000e000 1.0000000e000\n",                    # two coordinates
    b"v 1.0000000e000 1.0000000e000 1.0000000e000\nf 1 1\n",       # two indices
    b"v 1.0000000e000 1.0000000e000 1.0000000e000\nf 1 1 2\n",     # index past the last vertex
    b"v 1.0000000e000 1.0000000e000 1.0000000e000\nf 0 1 1\n",     # index 0 (1-based format)
    b"v 1.0000000e000 1.0000000e000 1.0000000e000\nf 1/1 1/1 1/1\n",
    b"v 1.0000000e000 1.0000000e000 1.0000000e000\nf 123456789 1 1\n",  # more than 8 digits
])
def test_loader_rejects_malformed_text(text):
    with pytest.raises(R.InvertError) as e:
        R.Mesh.parse_obj(text)
    assert e.value.code == api.ERR_FORMAT


def test_loader_edge_cases(tmp_path):
    m = R.Mesh.parse_obj(b"# only a comment\n")
    assert m.counts() == (0, 0)
    m = R.Mesh.parse_obj(b"f 1 2 3\nv 0.0000000e000 0.0000000e000 0.0000000e000\n"
                         b"v 1.0000000e000 0.0000000e000 0.0000000e000\nv -0.0000000e-000 9.9999999e002 1.0000000e-038\n")
    assert m.counts() == (3, 3) and list(m.indices) == [0, 1, 2]  # faces may precede vertices
    v = m.vertices
    assert np.signbit(v[2, 0]) and v[2, 0] == 0 and v[2, 1] == np.float32(999.99999)
    with pytest.raises(R.InvertError) as e:
        R.Mesh.load_obj(str(tmp_path / "missing.obj"))
    assert e.value.code == api.ERR_IO  # the reference would dereference a null FILE* (src/invert.cpp:44-45)


def test_mesh_from_arrays_checks():
    xyzw = np.zeros((3, 4), np.float32)
    m = R.Mesh.from_arrays(xyzw, np.array([0, 1, 2], np.uint32))
    assert m.counts() == (3, 3)
    with pytest.raises(R.InvertError):
        R.Mesh.from_arrays(xyzw, np.array([0, 1, 3], np.uint32))
    with pytest.raises(R.InvertError):
        R.Mesh.from_arrays(xyzw, np.array([0, 1], np.uint32))


@pytest.mark.parametrize("size", [(512, 512), (1920, 1080), (3840, 2160), (333, 77), (1, 1)])
def test_camera_is_bit_identical_to_the_oracle(oracle, size):
    assert bytes(R.camera_reference(*size)) == bytes(oracle.camera_reference(*size))


def test_camera_pins():
    p = common.pins()["cameras"]["1920x1080"]
    cam = R.camera_reference(1920, 1080)
    for name in ("model", "view", "proj", "cam_angle"):
        assert np.array_equal(np.array(getattr(cam, name), np.float32).view(np.uint32), common.unhex(p[name]).view(np.uint32))
    with pytest.raises(R.InvertError):
        R.camera_reference(0, 10)


def test_turntable_camera(oracle):
    """elapsed * 90 degrees about the up axis on the model matrix; bit-identical to the oracle's; elapsed = 0 is the
    reference camera; one second is a quarter turn (x -> -z)."""
    for t in (0.0, 0.3, 1.0, 2.5, -7.25, 1234.5):
        assert bytes(R.camera_turntable(1920, 1080, t)) == bytes(oracle.camera_turntable(1920, 1080, t))
    ref, zero = R.camera_reference(640, 480), R.camera_turntable(640, 480, 0.0)
    assert np.array_equal(np.array(zero.model, np.float32), np.array(ref.model, np.float32))
    assert bytes(zero)[64:] == bytes(ref)[64:]  # view, projection and camera angle are untouched
    m = np.array(R.camera_turntable(640, 480, 1.0).model, np.float32).reshape(4, 4).T  # column-major storage
    assert np.allclose(m @ np.array([1, 0, 0, 1], np.float32), [0, 0, -1, 1], atol=1e-6)
    with pytest.raises(R.InvertError):
        R.camera_turntable(640, 480, float("nan"))


def test_tile_layout():
    prm = R.make_params(1920, 1080)
    npix, ids = R.tile_layout(prm)
    assert npix == 1920 * 1080 and len(ids) == 30 * 17
    total = []
    for r in range(8):
        prm = R.make_params(1920, 1080, tile=64, rank=r, nranks=8)
        npix, ids = R.tile_layout(prm)
        assert npix == 64 * 64 * 64  # ceil(510 / 8) slots
        assert (ids % 8 == r).all()
        total += list(ids)
    assert sorted(total) == list(range(510))
    for bad in (dict(tile=12), dict(rank=3, nranks=2), dict(width=0), dict(tile=2048)):
        kw = dict(width=64, height=64)
        kw.update(bad)
        with pytest.raises(R.InvertError):
            R.tile_layout(R.make_params(**kw))
    # the largest frames: up to 2^31 - 1 pixels are laid out (64-bit pixel counts), more is refused
    npix, ids = R.tile_layout(R.make_params(46340, 46340))
    assert npix == 46340 * 46340 and len(ids) == 725 * 725
    npix, _ = R.tile_layout(R.make_params(65535, 32767, tile=1024, rank=5, nranks=7))
    assert npix == -(-(64 * 32) // 7) * 1024 * 1024
    for w, h in ((46341, 46341), (65536, 32768), (0xFFFFFFFF, 0xFFFFFFFF), (65537, 1), (1, 70000), (0x80000000, 1)):
        with pytest.raises(R.InvertError) as e:
            R.tile_layout(R.make_params(w, h))
        assert e.value.code == api.ERR_ARG
    # bounds on what the kernels loop over or divide by: samples per pixel, ranks (a rank count near 2^32 would wrap
    # the slot arithmetic to an empty frame)
    R.tile_layout(R.make_params(64, 64, spp=4096))
    R.tile_layout(R.make_params(64, 64, rank=65535, nranks=65536))
    for bad in (dict(spp=4097), dict(spp=0xFFFFFFFF), dict(nranks=65537), dict(rank=1, nranks=0xFFFFFFFF)):
        with pytest.raises(R.InvertError) as e:
            R.tile_layout(R.make_params(64, 64, **bad))
        assert e.value.code == api.ERR_ARG


def test_no_gpu_is_a_loud_error():
    """Without a device the library refuses to create a context; nothing falls back to the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(R.InvertError) as e:
        R.Context(0)
    assert e.value.code == api.ERR_NO_DEVICE
    mesh = R.Mesh.from_arrays(np.zeros((3, 4), np.float32), np.array([0, 1, 2], np.uint32))
    with pytest.raises(R.InvertError) as e:
        mesh.pin()  # page-locked memory needs a device; the mesh stays usable as it is
    assert e.value.code == api.ERR_NO_DEVICE
    assert mesh.counts() == (3, 3) and mesh.indices.tolist() == [0, 1, 2]


def test_product_does_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "renderer_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h", ".inc")) or fn == "Makefile":
                text = open(os.path.join(dirpath, fn), errors="replace").read()
                assert "liboracle" not in text and "oracle/" not in text and "import oracle" not in text, fn


def test_general_obj_reader(tmp_path, bunny_path, bunny_mesh):
    """Tolerant reader: same topology as the fixed-format loader on the bunny, coordinates = correctly rounded text."""
    g = R.Mesh.load_obj_general(bunny_path)
    xyzw, idx = bunny_mesh
    assert g.counts() == (2503, 14904) and np.array_equal(g.indices, idx)
    lines = [ln for ln in objfmt.bunny().text().decode().splitlines() if ln.startswith("v ")]
    exact = np.array([[np.float32(float(t)) for t in ln.split()[1:]] for ln in lines], np.float32)
    assert np.array_equal(g.vertices[:, :3].view(np.uint32), exact.view(np.uint32))
    ulp = np.abs(g.vertices[:, :3].view(np.int32).astype(np.int64) - xyzw[:, :3].view(np.int32))
    assert int(ulp.max()) == 3  # the reference loader's own arithmetic is up to 3 ulp away (src/invert.cpp:139-152)
    text = (b"# comment\r\nmtllib x.mtl\r\no thing\r\nv 0 0 0\r\nv 1.5 0 0 1.0\r\nv 1 1e0 -2.5E-1\r\nv  0\t1 0\r\n"
            b"vn 0 0 1\r\nvt 0.5 0.5\r\ns off\r\nf 1/1/1 2/1/1 3/1/1 4/1/1\r\nf -4//1 -3//1 -1//1\r\nf 1 3 4")  # no final newline
    path = tmp_path / "general.obj"
    path.write_bytes(text)
    m = R.Mesh.load_obj_general(str(path))
    assert m.counts() == (4, 12)
    assert m.indices.tolist() == [0, 1, 2, 0, 2, 3, 0, 1, 3, 0, 2, 3]  # quad fanned, relative indices resolved
    assert m.vertices[:, :3].tolist() == [[0, 0, 0], [1.5, 0, 0], [1, 1, -0.25], [0, 1, 0]]
    for bad in (b"v 1 2\n", b"v 0 0 0\nf 1 2 3\n", b"v 0 0 0\nf 1 1\n", b"v 0 0 0\nf 0 1 1\n", b"v a b c\n"):
        path.write_bytes(bad)
        with pytest.raises(R.InvertError) as e:
            R.Mesh.load_obj_general(str(path))
        assert e.value.code == api.ERR_FORMAT
    with pytest.raises(R.InvertError) as e:
        R.Mesh.load_obj_general(str(tmp_path / "missing.obj"))
    assert e.value.code == api.ERR_IO


def test_image_writers(tmp_path):
    color, depth, primid = common.golden_frame("bunny_512")
    api.write_ppm(str(tmp_path / "c.ppm"), color)
    api.write_pfm(str(tmp_path / "d.pfm"), depth)
    api.write_primid(str(tmp_path / "p.bin"), primid)
    raw = (tmp_path / "c.ppm").read_bytes()
    head = b"P6\n512 512\n255\n"
    assert raw.startswith(head) and len(raw) == len(head) + 512 * 512 * 3
    rgb = np.frombuffer(raw[len(head):], np.uint8).reshape(512, 512, 3)
    assert np.array_equal(rgb, color[..., [2, 1, 0]])  # B,G,R,A in memory -> R,G,B in the file
    api.write_ppm(str(tmp_path / "c2.ppm"), color[..., [2, 1, 0, 3]], flags=R.RGBA)
    assert (tmp_path / "c2.ppm").read_bytes() == raw
    # PNG: decoded by an independent reader (Pillow), byte for byte R,G,B,A; odd extents and more than one stored block
    from PIL import Image
    api.write_png(str(tmp_path / "c.png"), color)
    img = Image.open(str(tmp_path / "c.png"))
    assert img.mode == "RGBA" and img.size == (512, 512) and "srgb" in img.info
    assert np.array_equal(np.asarray(img), color[..., [2, 1, 0, 3]])
    small = np.arange(7 * 5 * 4, dtype=np.uint8).reshape(5, 7, 4)
    api.write_png(str(tmp_path / "s.png"), small, flags=R.RGBA)
    assert np.array_equal(np.asarray(Image.open(str(tmp_path / "s.png"))), small)
    raw = (tmp_path / "d.pfm").read_bytes()
    head = b"Pf\n512 512\n-1.0\n"
    assert raw.startswith(head)
    assert np.array_equal(np.frombuffer(raw[len(head):], "<f4").reshape(512, 512)[::-1], depth)  # bottom row first
    raw = (tmp_path / "p.bin").read_bytes()
    assert raw[:4] == b"PID1" and np.frombuffer(raw[4:16], "<u4").tolist() == [512, 512, 0]
    assert np.array_equal(np.frombuffer(raw[16:], "<u4").reshape(512, 512), primid)
    with pytest.raises(R.InvertError) as e:
        api.write_ppm(str(tmp_path / "nodir" / "x.ppm"), color)
    assert e.value.code == api.ERR_IO


def test_division_by_frame_constants_is_exact(tmp_path):
    """make_ray divides by the extent and by the projection terms with multiply-adds on a host-rounded reciprocal
    (div_by, rt_core.h); tools/micro/div_check.c holds the same statements and compares them with IEEE division for
    every float32 numerator of a range: here a slice of each range a ray uses, the whole ranges in the tool's header."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "micro", "div_check.c")
    exe = str(tmp_path / "div_check")
    subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-fopenmp", "-o", exe, src, "-lm"])
    # pixel coordinates over extents; normalised coordinates over projection terms (reference camera: 45 degrees)
    for lo, hi, divisors in (("1", "64", ["1920", "1080", "3840", "2160", "512", "333", "46340"]),
                             ("0.25", "1", ["2.41421342", "1.35799515", "1.81066012", "0.0087268678", "114591.56"])):
        p = subprocess.run([exe, lo, hi] + divisors, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout + p.stderr
        assert p.stdout.count("0 quotients differ") == len(divisors)
