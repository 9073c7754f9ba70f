"""CPU tests: the oracle against the facts that pin it.

The reference has no tests, goldens or known-answer vectors (parity unpinned), so
the pins are (a) facts about res/bunny.obj and the reference's literals that can be
derived independently of the oracle's code, and (b) this repo's committed goldens.
"""
import hashlib
import sys

import numpy as np
import pytest

import common
import objfmt


def test_bunny_text_regenerates_byte_for_byte():
    text = objfmt.bunny().text()
    assert len(text) == 198443
    assert hashlib.sha256(text).hexdigest() == objfmt.bunny_sha256()
    assert objfmt.bunny_sha256().startswith("c93e4a6f") and objfmt.bunny_sha256().endswith("0d55ef")


def test_loader_counts_and_first_vertices(bunny_mesh):
    xyzw, idx = bunny_mesh
    assert xyzw.shape == (2503, 4) and idx.size == 14904  # res/bunny.obj:2-3, src/invert.cpp:77-79
    assert idx.min() == 0 and idx.max() == 2502 and np.unique(idx).size == 2503
    want = [("-0x1.befaa6p-9", "0x1.0ae4fep-3", "0x1.646c70p-6"),
            ("-0x1.4eb8c0p-4", "0x1.3852aep-3", "0x1.e5e2a8p-6"),
            ("-0x1.f46ca4p-6", "0x1.ff181ep-4", "0x1.1fec58p-10")]
    for i, row in enumerate(want):
        assert [float(v) for v in xyzw[i, :3]] == [float.fromhex(s) for s in row]


def test_loader_arithmetic_is_not_strtof(bunny_mesh):
    """src/invert.cpp:139-152 rounds differently from a correctly rounded parse on 2844 of 7509 coordinates."""
    xyzw, _ = bunny_mesh
    lines = [ln for ln in objfmt.bunny().text().decode().splitlines() if ln.startswith("v ")]
    exact = np.array([[np.float32(float(t)) for t in ln.split()[1:]] for ln in lines], np.float32)
    ulp = np.abs(xyzw[:, :3].view(np.int32).astype(np.int64) - exact.view(np.int32))
    assert int((ulp != 0).sum()) == 2844 and int(ulp.max()) == 3
    # 10^-3 by repeated division is one ulp below the correctly rounded constant
    n = np.float32(1.0)
    for _ in range(3):
        n = np.float32(n / np.float32(10.0))
    assert float(n).hex() == "0x1.0624dc0000000p-10"


def test_loader_restatement_in_numpy(bunny_mesh):
    """Independent numpy float32 restatement of src/invert.cpp:139-152 from the decimal fields."""
    xyzw, idx = bunny_mesh
    fo = objfmt.bunny()
    whole = (fo.mant // 10_000_000).astype(np.float32)
    frac = (fo.mant % 10_000_000).astype(np.float32)
    norm = np.ones(fo.mant.shape, np.float32)
    for step in range(int(fo.expo.max())):
        active = fo.expo > step
        norm = np.where(active & fo.eneg, (norm / np.float32(10.0)).astype(np.float32), norm)
        norm = np.where(active & ~fo.eneg, (norm * np.float32(10.0)).astype(np.float32), norm)
    val = ((whole + (frac / np.float32(10000000.0)).astype(np.float32)).astype(np.float32) * norm).astype(np.float32)
    val = (val * np.where(fo.neg, np.float32(-1.0), np.float32(1.0))).astype(np.float32)
    assert np.array_equal(val.view(np.uint32), xyzw[:, :3].view(np.uint32))
    assert np.array_equal(idx.reshape(-1, 3), fo.faces - 1)


def test_loader_pins(bunny_mesh):
    xyzw, idx = bunny_mesh
    p = common.pins()
    assert hashlib.sha256(np.ascontiguousarray(xyzw[:, :3]).tobytes()).hexdigest() == p["vertices_sha256"]
    assert hashlib.sha256(idx.tobytes()).hexdigest() == p["indices_sha256"]
    assert np.allclose(xyzw[:, :3].min(0), [-0.09438042, 0.0333099, -0.06167917], atol=1e-8)
    assert np.allclose(xyzw[:, :3].max(0), [0.0607788, 0.18699603, 0.05871464], atol=1e-8)


def test_mesh_topology(bunny_mesh):
    xyzw, idx = bunny_mesh
    f = idx.reshape(-1, 3).astype(np.int64)
    e = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), axis=1)
    uniq, counts = np.unique(e, axis=0, return_counts=True)
    assert len(uniq) == 7473 and int((counts == 1).sum()) == 42 and int((counts > 2).sum()) == 0
    v = xyzw[:, :3].astype(np.float64)
    vol = np.einsum("ij,ij->i", v[f[:, 0]], np.cross(v[f[:, 1]], v[f[:, 2]])).sum() / 6.0
    assert abs(vol - 7.63e-4) < 2e-6  # outward CCW winding


def test_camera_matches_literals(oracle):
    cam = oracle.camera_reference(1920, 1080)
    view = np.array(cam.view, np.float32).reshape(4, 4)  # [col][row]
    s = np.float32(np.sin(np.float64(np.float32(180.0) * np.float32(0.017453292519943295))))
    assert float(s).hex() == "-0x1.777a5c0000000p-24"  # sin(pi_f32) != 0 leaks into the view matrix
    assert view[0, 0] == -1 and view[1, 1] == -1 and view[2, 2] == 1 and view[3, 2] == -1 and view[3, 3] == 1
    assert view[0, 1] == s and view[1, 0] == -s
    proj = np.array(cam.proj, np.float32).reshape(4, 4)
    t = np.tan(np.float64(np.float32(np.float32(45.0) * np.float32(0.017453292519943295)) / np.float32(2.0)))
    th = np.float32(t)
    assert proj[1, 1] == np.float32(1.0) / th
    assert proj[0, 0] == np.float32(1.0) / (np.float32(np.float32(1920.0) / np.float32(1080.0)) * th)
    assert proj[2, 2] == np.float32(10.0) / (np.float32(0.1) - np.float32(10.0)) and proj[2, 3] == -1
    assert proj[3, 2] == -(np.float32(10.0) * np.float32(0.1)) / (np.float32(10.0) - np.float32(0.1))
    assert list(cam.cam_angle) == [0.0, 0.0, -1.0]  # src/invert.cpp:633


@pytest.mark.parametrize("size", ["512x512", "1920x1080", "3840x2160"])
def test_camera_pins(oracle, bunny_mesh, size):
    w, h = [int(v) for v in size.split("x")]
    cam = oracle.camera_reference(w, h)
    p = common.pins()["cameras"][size]
    for name in ("model", "view", "proj", "cam_angle"):
        assert np.array_equal(np.array(getattr(cam, name), np.float32).view(np.uint32), common.unhex(p[name]).view(np.uint32))
    frame = oracle.frame_setup(bunny_mesh[0], cam)
    assert np.array_equal(frame.view(np.uint32), common.unhex(common.pins()["frames"][size]).view(np.uint32))


def test_vertex_shader_matches_numpy(oracle, bunny_mesh):
    xyzw, _ = bunny_mesh
    cam = oracle.camera_reference(1920, 1080)
    clip, col = oracle.vertex_shader(xyzw, cam)
    p = xyzw.copy()
    p[:, 3] = 1.0

    def mv(m, v):  # column-major, left-to-right float32 sum (src/shader.slang:25)
        m = np.array(m, np.float32).reshape(4, 4)
        out = np.zeros_like(v)
        for r in range(4):
            acc = (m[0, r] * v[:, 0]).astype(np.float32)
            for c in range(1, 4):
                acc = (acc + (m[c, r] * v[:, c]).astype(np.float32)).astype(np.float32)
            out[:, r] = acc
        return out

    want = mv(cam.proj, mv(cam.view, mv(cam.model, p)))
    assert np.array_equal(want.view(np.uint32), clip.view(np.uint32))
    assert np.array_equal(col[:, 2], (xyzw[:, 2] * np.float32(10.0)).astype(np.float32)) and (col[:, :2] == 1).all()
    zv = -clip[:, 3]
    assert zv.min() > -1.0625 and zv.max() < -0.94  # no clipping against near 0.1 / far 10


@pytest.mark.parametrize("size,covered", [((512, 512), 5651), ((1920, 1080), 25139)])
def test_rasteriser_facts(oracle, bunny_mesh, size, covered):
    """A3+A4 as a double-precision rasteriser: 2565 front faces, pixel counts, screen rectangle."""
    xyzw, idx = bunny_mesh
    cam = oracle.camera_reference(*size)
    depth, primid, nfront = oracle.raster(xyzw, idx, cam, *size)
    hit = primid != 0xFFFFFFFF
    assert nfront == 2565 and int(hit.sum()) == covered
    assert 0.90279 <= depth[hit].min() and depth[hit].max() <= 0.91496
    if size == (1920, 1080):
        ys, xs = np.nonzero(hit)
        assert (xs.min(), xs.max(), ys.min(), ys.max()) == (880, 1085, 301, 496)


def test_ray_casting_reproduces_the_rasteriser(oracle, bunny_mesh):
    """Primary rays + Moller-Trumbore pick the same primitive as the rasteriser at every pixel of config 1."""
    xyzw, idx = bunny_mesh
    cam = oracle.camera_reference(512, 512)
    color, depth, primid = oracle.render_bruteforce(xyzw, idx, cam, oracle.make_params(512, 512))
    rdepth, rprim, _ = oracle.raster(xyzw, idx, cam, 512, 512)
    assert int((primid != rprim).sum()) <= 2  # pixel centres within rounding of an edge may differ
    same = (primid == rprim) & (primid != 0xFFFFFFFF)
    assert np.abs(depth[same].astype(np.float64) - rdepth[same]).max() < 8 * 2.0 ** -24
    hit = primid != 0xFFFFFFFF
    assert (color[..., 1][hit] == 255).all() and (color[..., 2][hit] == 255).all() and (color[..., 3] == 255).all()
    assert color[..., 0][hit].max() == 201 and (color[~hit] == [0, 0, 0, 255]).all()
    z10 = np.clip(10.0 * xyzw[:, 2].astype(np.float64), 0, 1)
    assert color[..., 0][hit].max() <= oracle.srgb8(z10.max())


@pytest.mark.parametrize("name", ["bunny_512", "bunny_1080_rect"])
def test_oracle_reproduces_goldens(oracle, bunny_mesh, name):
    xyzw, idx = bunny_mesh
    want = common.golden_frame(name)
    h, w = want[1].shape
    cam = oracle.camera_reference(w, h)
    got = oracle.render_bvh(xyzw, idx, cam, oracle.make_params(w, h))
    common.assert_frames_equal(got, want, name)
    p = common.pins()[name]
    assert int((got[2] != 0xFFFFFFFF).sum()) == p["hits"]
    assert hashlib.sha256(got[2].tobytes()).hexdigest() == p["primid_sha256"]
    assert hashlib.sha256(got[1].tobytes()).hexdigest() == p["depth_sha256"]
    assert hashlib.sha256(got[0].tobytes()).hexdigest() == p["color_sha256"]


@pytest.mark.parametrize("flags", [0, 1, 2, 3, 7])
def test_bvh_oracle_equals_brute_force(oracle, bunny_mesh, flags):
    xyzw, idx = bunny_mesh
    cam = oracle.camera_reference(256, 192)
    prm = oracle.make_params(256, 192, flags=flags)
    common.assert_frames_equal(oracle.render_bvh(xyzw, idx, cam, prm), oracle.render_bruteforce(xyzw, idx, cam, prm))


def test_bvh_oracle_equals_brute_force_jittered(oracle, bunny_mesh):
    xyzw, idx = bunny_mesh
    cam = oracle.camera_reference(160, 120)
    prm = oracle.make_params(160, 120, spp=4, seed=7, flags=1)
    common.assert_frames_equal(oracle.render_bvh(xyzw, idx, cam, prm), oracle.render_bruteforce(xyzw, idx, cam, prm))


def test_bvh_oracle_on_random_soup(oracle):
    xyz, faces = objfmt.random_soup(700, seed=3)
    xyzw, idx = oracle.parse_obj(objfmt.from_floats(xyz, faces).text())
    cam = oracle.camera_reference(200, 200)
    prm = oracle.make_params(200, 200)
    a = oracle.render_bvh(xyzw, idx, cam, prm)
    b = oracle.render_bruteforce(xyzw, idx, cam, prm)
    common.assert_frames_equal(a, b)
    assert (a[2] != 0xFFFFFFFF).sum() > 1000


def test_srgb_table(oracle):
    """Thresholds against a float64 evaluation of the sRGB OETF (src/invert.cpp:519 stores B8G8R8A8_SRGB)."""
    t = np.load(common.GOLDEN + "/srgb_thresholds.npy")

    def enc(x):
        x = np.float64(x)
        s = 12.92 * x if x <= 0.0031308 else 1.055 * x ** (1 / 2.4) - 0.055
        return int(np.floor(255.0 * s + 0.5))

    for k in range(1, 256):
        assert enc(t[k]) == k and enc(np.nextafter(t[k], np.float32(-1))) == k - 1
        assert oracle.srgb8(t[k]) == k and oracle.srgb8(np.nextafter(t[k], np.float32(-1))) == k - 1
    assert oracle.srgb8(-1.0) == 0 and oracle.srgb8(float("nan")) == 0 and oracle.srgb8(2.0) == 255
    assert oracle.srgb8(0.0) == 0 and oracle.srgb8(1.0) == 255


def test_sample_offsets(oracle):
    assert oracle.sample_offset(12345, 0, 99) == (0.5, 0.5)
    pts = np.array([oracle.sample_offset(p, s, 1) for p in range(64) for s in range(1, 17)])
    assert pts.min() >= 0.0 and pts.max() < 1.0
    assert abs(pts.mean() - 0.5) < 0.03 and len(np.unique(pts[:, 0])) > 1000
    assert oracle.sample_offset(5, 3, 1) != oracle.sample_offset(5, 3, 2)


def test_occlusion_oracle_bvh_equals_brute_force(oracle, bunny_mesh):
    xyzw, idx = bunny_mesh
    cam = oracle.camera_reference(120, 90)
    prm = oracle.make_params(120, 90, ao_rays=12, seed=3, flags=1)
    a = oracle.render_bvh(xyzw, idx, cam, prm)
    b = oracle.render_bruteforce(xyzw, idx, cam, prm)
    common.assert_frames_equal(a, b, "ao")
    plain = oracle.render_bvh(xyzw, idx, cam, oracle.make_params(120, 90, flags=1))
    hit = a[2] != 0xFFFFFFFF
    assert np.array_equal(a[1], plain[1]) and np.array_equal(a[2], plain[2])  # only colour changes
    assert (a[0][hit].astype(int) <= plain[0][hit].astype(int)).all() and (a[0][hit] < plain[0][hit]).any()
    # an open surface facing the camera with nothing around it is not occluded at all
    tri = np.array([[-0.05, 0.05, 0.0], [0.05, 0.05, 0.0], [0.0, 0.15, 0.0]])
    txyzw, tidx = oracle.parse_obj(__import__("objfmt").from_floats(tri, np.array([[0, 1, 2]], np.uint32)).text())
    c = oracle.camera_reference(64, 64)
    with_ao = oracle.render_bruteforce(txyzw, tidx, c, oracle.make_params(64, 64, ao_rays=32))
    without = oracle.render_bruteforce(txyzw, tidx, c, oracle.make_params(64, 64))
    assert np.array_equal(with_ao[0], without[0]) and (with_ao[2] != 0xFFFFFFFF).sum() > 20


def _bunny_from_the_decimal_fields():
    """The mesh without the oracle: the loader's float32 arithmetic restated in numpy (test_loader_restatement_in_numpy)."""
    fo = objfmt.bunny()
    whole = (fo.mant // 10_000_000).astype(np.float32)
    frac = (fo.mant % 10_000_000).astype(np.float32)
    norm = np.ones(fo.mant.shape, np.float32)
    for step in range(int(fo.expo.max())):
        active = fo.expo > step
        norm = np.where(active & fo.eneg, (norm / np.float32(10.0)).astype(np.float32), norm)
        norm = np.where(active & ~fo.eneg, (norm * np.float32(10.0)).astype(np.float32), norm)
    val = ((whole + (frac / np.float32(10000000.0)).astype(np.float32)).astype(np.float32) * norm).astype(np.float32)
    val = (val * np.where(fo.neg, np.float32(-1.0), np.float32(1.0))).astype(np.float32)
    return val, (fo.faces - 1).astype(np.int64)


@pytest.mark.parametrize("size,golden,covered", [("512x512", "bunny_512", 5651), ("1920x1080", "bunny_1080_rect", 25139)])
def test_goldens_against_the_numpy_rasteriser(size, golden, covered):
    """Every pixel of configs 1 and 2: primitive id, depth and the stored colour bytes of the committed goldens
    against a double-precision rasteriser that shares no code with the oracle (tests/raster_np.py), fed with the
    mesh from the decimal fields and the pinned matrices.  Primitive ids agree except where a pixel centre lies
    within rounding of an edge; depth within a few float32 roundings (99.9 % of the pixels within 4); blue within 1 LSB (8-bit sRGB); red, green and
    alpha 255; background (0, 0, 0, 255) in B, G, R, A order."""
    import raster_np
    w, h = [int(v) for v in size.split("x")]
    xyz, faces = _bunny_from_the_decimal_fields()
    cam = common.pins()["cameras"][size]
    primid, depth, blue, _, nfront = raster_np.rasterise(xyz, faces, common.unhex(cam["model"]), common.unhex(cam["view"]),
                                                         common.unhex(cam["proj"]), w, h)
    assert nfront == 2565 and int((primid != 0xFFFFFFFF).sum()) == covered
    gc, gd, gp = common.golden_frame(golden)
    differ = gp != primid
    assert int(differ.sum()) <= 4
    same = ~differ & (gp != 0xFFFFFFFF)
    # (float32 ray casting against double-precision interpolation: a few roundings, more on a triangle seen nearly
    # edge-on, where the ray parameter is a quotient of two small numbers: one pixel of the 25 139 is 18 units off)
    err = np.abs(gd[same].astype(np.float64) - depth[same]) / 2.0 ** -24
    assert np.percentile(err, 99.9) < 4.0 and err.max() < 32.0
    want_b = raster_np.srgb8(blue)
    assert np.abs(gc[..., 0][same].astype(np.int64) - want_b[same]).max() <= 1
    assert (gc[..., 0][same].astype(np.int64) == want_b[same]).mean() > 0.97  # off by one only at rounding boundaries
    hit = gp != 0xFFFFFFFF
    assert (gc[..., 1][hit] == 255).all() and (gc[..., 2][hit] == 255).all() and (gc[..., 3] == 255).all()
    assert (gc[~hit] == [0, 0, 0, 255]).all() and (gd[~hit] == 1.0).all()


def test_lambert_and_invert_against_numpy(oracle, bunny_mesh):
    """The build-defined shading on top of the reference's colour: (1, 1, 10 z) times max(0, n . -cameraAngle) with the
    triangle's geometric normal (cameraAngle = (0, 0, -1), src/invert.cpp:633), sRGB-encoded, then 255 - c on R, G, B:
    the oracle's frame against the numpy rasteriser's interpolated colour and a float64 evaluation, within 1 LSB."""
    import raster_np
    w, h = 640, 360
    xyz, faces = _bunny_from_the_decimal_fields()
    cam = oracle.camera_reference(w, h)
    primid, _, blue, lam, _ = raster_np.rasterise(xyz, faces, cam.model, cam.view, cam.proj, w, h, light=(0.0, 0.0, 1.0))
    for flags in (1, 3, 2):  # Lambert, Lambert + invert, invert alone
        color, _, op = oracle.render_bvh(*bunny_mesh, cam, oracle.make_params(w, h, flags=flags))
        same = (op == primid) & (op != 0xFFFFFFFF)
        assert int((op != primid).sum()) <= 4 and same.sum() > 2500
        L = lam if flags & 1 else np.ones_like(lam)
        want = np.stack([raster_np.srgb8(np.clip(blue, 0.0, 1.0) * L), raster_np.srgb8(L), raster_np.srgb8(L)], axis=-1)  # B, G, R
        if flags & 2:
            want = 255 - want
        diff = np.abs(color[..., :3][same].astype(np.int64) - want[same])
        assert diff.max() <= 1 and (diff == 0).mean() > 0.97
        assert (color[..., 3] == 255).all()
        bg = 255 if flags & 2 else 0
        assert (color[..., :3][op == 0xFFFFFFFF] == bg).all()


@pytest.mark.parametrize("size,golden", [("512x512", "bunny_512"), ("1920x1080", "bunny_1080_rect")])
def test_vulkan_raster_rules_bound_the_unpinned_part(size, golden):
    """The goldens are ray casts through pixel centres with inclusive edges and unsnapped positions; what
    vkCmdDrawIndexed (src/invert.cpp:828; state :1331-1337, :1375-1385) draws is decided by a rasteriser with the
    top-left fill rule and positions snapped to subPixelPrecisionBits.  The numpy rasteriser under those rules against
    the goldens, per pixel: the top-left rule alone moves nothing (no pixel centre lies exactly on an edge); snapping
    to 8 sub-pixel bits hands 5 of 5 651 / 13 of 25 139 covered pixels to a neighbouring triangle and changes no
    pixel between hit and miss; the 4 bits the specification allows as a minimum move 138 / 275 (5 / 6 of them between
    hit and miss).  The counts and the pixels are pinned in pins.json (tests/golden/make_raster_bounds.py)."""
    sys.path.insert(0, common.GOLDEN)
    import make_raster_bounds as B
    pinned = common.pins()["vulkan_raster_bounds"][size]
    assert set(pinned) == set(B.VARIANTS)
    for variant in B.VARIANTS:
        got = B.bounds(size, golden, variant)
        assert got == pinned[variant], variant
    n = pinned["inclusive_edges_unsnapped"]["golden_covered"]
    assert pinned["top_left_unsnapped"]["primitive_id_differs"] == pinned["inclusive_edges_unsnapped"]["primitive_id_differs"] <= 1
    eight, four = pinned["top_left_8_subpixel_bits"], pinned["top_left_4_subpixel_bits"]
    assert eight["primitive_id_differs"] <= 0.001 * n and eight["hit_in_golden_only"] == eight["hit_in_variant_only"] == 0
    assert four["primitive_id_differs"] <= 0.03 * n and four["hit_in_golden_only"] + four["hit_in_variant_only"] <= 6
    # a neighbour's depth at the same pixel: the snapped variants stay within 2^-13 of the goldens where the ids agree
    assert eight["max_depth_difference_in_2^-24"] < 1024 and four["max_depth_difference_in_2^-24"] < 4096
