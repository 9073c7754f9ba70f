"""Shared helpers for the parity tests."""
import json
import os

import numpy as np

import objfmt

GOLDEN = objfmt.GOLDEN_DIR


def pins():
    with open(os.path.join(GOLDEN, "pins.json")) as f:
        return json.load(f)


def unhex(values):
    return np.array([float.fromhex(v) for v in values], np.float32)


def golden_frame(name, flags_invert=False):
    """Expand a cropped golden to full-frame (colour, depth, primid) arrays."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    w, h = [int(v) for v in z["size"]]
    x0, y0, x1, y1 = [int(v) for v in z["rect"]]
    color = np.zeros((h, w, 4), np.uint8)
    color[..., 3] = 255
    depth = np.ones((h, w), np.float32)
    primid = np.full((h, w), 0xFFFFFFFF, np.uint32)
    color[y0:y1, x0:x1] = z["color"]
    depth[y0:y1, x0:x1] = z["depth"]
    primid[y0:y1, x0:x1] = z["primid"]
    return color, depth, primid


def assert_frames_equal(got, want, what=""):
    """Bit-exact on primitive id, depth bits and colour bytes (integer / byte work: no tolerance)."""
    gc, gd, gp = got
    wc, wd, wp = want
    bad_p = np.argwhere(gp != wp)
    assert bad_p.size == 0, "%s primitive id differs at %d pixels, first (y,x)=%s got %s want %s" % (
        what, len(bad_p), bad_p[0], gp[tuple(bad_p[0])], wp[tuple(bad_p[0])])
    bad_d = np.argwhere(gd.view(np.uint32) != wd.view(np.uint32))
    assert bad_d.size == 0, "%s depth bits differ at %d pixels, first (y,x)=%s got %r want %r" % (
        what, len(bad_d), bad_d[0], gd[tuple(bad_d[0])], wd[tuple(bad_d[0])])
    bad_c = np.argwhere((gc != wc).any(axis=-1))
    assert bad_c.size == 0, "%s colour differs at %d pixels, first (y,x)=%s got %s want %s" % (
        what, len(bad_c), bad_c[0], gc[tuple(bad_c[0])], wc[tuple(bad_c[0])])


def slots_to_frame(slot_arrays, prm_w, prm_h, tile, nranks):
    """Host restatement of the tile de-interleave: per-rank [slots,tile,tile(,c)] arrays -> one frame."""
    tiles_x = (prm_w + tile - 1) // tile
    tiles_y = (prm_h + tile - 1) // tile
    first = slot_arrays[0]
    frame = np.zeros((tiles_y * tile, tiles_x * tile) + first.shape[3:], first.dtype)
    for tid in range(tiles_x * tiles_y):
        r, slot = tid % nranks, tid // nranks
        ty, tx = divmod(tid, tiles_x)
        frame[ty * tile:(ty + 1) * tile, tx * tile:(tx + 1) * tile] = slot_arrays[r][slot]
    return frame[:prm_h, :prm_w]


def grid_translations(n_side, spacing=0.2):
    """n_side x n_side grid of instance translations in the xy plane, centred on the origin (float32)."""
    c = (np.arange(n_side, dtype=np.float32) - np.float32((n_side - 1) / 2.0)) * np.float32(spacing)
    gx, gy = np.meshgrid(c, c, indexing="xy")
    return np.stack([gx.reshape(-1), gy.reshape(-1), np.zeros(n_side * n_side, np.float32)], axis=1).astype(np.float32)


def flatten_instances(xyzw, idx, translations):
    """What inv_scene_build_instanced assembles: translated copies (float32 adds), primitive id = inst*ntri + k."""
    nv = xyzw.shape[0]
    out_v = np.zeros((len(translations) * nv, 4), np.float32)
    out_i = np.zeros(len(translations) * idx.size, np.uint32)
    for j, t in enumerate(translations.astype(np.float32)):
        out_v[j * nv:(j + 1) * nv, :3] = (xyzw[:, :3] + t[None, :]).astype(np.float32)
        out_i[j * idx.size:(j + 1) * idx.size] = idx + np.uint32(j * nv)
    return out_v, out_i


def pulled_back_camera(cam, distance, height=0.11, far=None):
    """The reference camera moved back along +z so that a grid of instances fits the view (same roll and lens).

    `far`: another far plane than the reference's 10 (src/invert.cpp:639) for grids that lie beyond it: the two terms
    of glm::perspective (zero-to-one depth) that hold it, in float32: P22 = far / (near - far), P32 = -(far near) /
    (far - near), near 0.1 as in the reference."""
    cam.view[12], cam.view[13], cam.view[14] = 0.0, float(np.float32(height)), float(-np.float32(distance))
    if far is not None:
        near, far = np.float32(0.1), np.float32(far)
        cam.proj[10] = float(np.float32(far / np.float32(near - far)))
        cam.proj[14] = float(np.float32(-np.float32(far * near) / np.float32(far - near)))
    return cam


def grid_camera(cam, side):
    """The camera the benchmarks and tests look at a side x side grid of instances with: pulled back by 0.27 per
    instance of the grid's side, and with the far plane at 100 once that is beyond the reference's far plane."""
    distance = 0.27 * side
    return pulled_back_camera(cam, distance, far=100.0 if distance + 0.5 > 10.0 else None)


def oracle_params(oracle, prm):
    """The oracle's params for a draw: the same frame without the flags that only choose a GPU kernel."""
    import renderer_amd as R
    return oracle.make_params(prm.width, prm.height, prm.spp, prm.ao_rays,
                              prm.flags & ~(R.BRUTE_FORCE | R.LANE_TRAVERSAL | R.BLOCKS | R.TILE_TRAVERSAL), prm.seed, ao_tmax=prm.ao_tmax)


def planes_to_frame(planes, w, h):
    """[3, h*w] int32 planes (colour, depth bits, primitive id) on the host -> (colour u8, depth f32, primid u32)."""
    return (planes[0].view(np.uint8).reshape(h, w, 4), planes[1].view(np.float32).reshape(h, w),
            planes[2].view(np.uint32).reshape(h, w))


def sparse_composite_single_gpu(scene, ctx, cam, w, h, nranks, tile, **kw):
    """The sparse composite with the ranks played one after another on this GPU: rank 0 renders into the frame,
    the other ranks' slot buffers are laid out as the gather would deliver them, inv_deinterleave_sparse finishes."""
    import torch

    import renderer_amd as R
    flags = kw.pop("flags", 0) | R.SPARSE
    rect = scene.hit_rect(cam, w, h)
    prm = [R.make_params(w, h, flags=flags, tile=tile, rank=r, nranks=nranks, **kw) for r in range(nranks)]
    grid, n, _ = R.sparse_layout(prm[1], rect)
    frame = torch.full((3, h * w), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    gathered = torch.full((nranks, 3, max(n, 1)), 0x3C3C3C3C, dtype=torch.int32, device="cuda")
    for r in range(nranks):
        o = frame if r == 0 else gathered[r]
        scene.render_device(cam, prm[r], o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
    ctx.deinterleave_sparse(prm[0], rect, gathered.data_ptr(), 3 * max(n, 1), max(n, 1), 3, frame.data_ptr(), h * w)
    ctx.synchronize()
    return planes_to_frame(frame.cpu().numpy(), w, h), rect, grid, n


def dense_composite_single_gpu(scene, ctx, cam, w, h, nranks, tile, **kw):
    """The dense composite with the ranks played one after another on this GPU: every rank renders its tile slots
    into its part of the gather buffer, inv_deinterleave rebuilds the three row-major planes."""
    import torch

    import renderer_amd as R
    prm = [R.make_params(w, h, tile=tile, rank=r, nranks=nranks, **kw) for r in range(nranks)]
    npix, _ = R.tile_layout(prm[-1])  # (rank 0 holds nothing when it is idle)
    gathered = torch.full((nranks, 3, npix), 0x3C3C3C3C, dtype=torch.int32, device="cuda")
    frame = torch.full((3, h * w), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    for r in range(nranks):
        o = gathered[r]
        scene.render_device(cam, prm[r], o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
    for k in range(3):
        ctx.deinterleave(prm[0], gathered[0, k].data_ptr(), frame[k].data_ptr(), rank_stride=3 * npix)
    ctx.synchronize()
    return planes_to_frame(frame.cpu().numpy(), w, h)


def check_lbvh(oracle, scene, xyzw, idx):
    """The GPU build's sorted keys, order, hierarchy and boxes against the oracle's sequential build of the same mesh."""
    got = scene.read_lbvh()
    want = oracle.lbvh_build(xyzw, idx)
    assert np.array_equal(got["morton"], want["morton"]), "sorted Morton keys"
    assert np.array_equal(got["order"], want["order"]), "sorted order (stable: ties keep primitive order)"
    assert np.array_equal(got["child"], want["child"]), "Karras hierarchy"
    assert np.array_equal(got["parent"], want["parent"])
    assert np.array_equal(got["leaf_parent"], want["leaf_parent"])
    assert np.array_equal(got["box"], want["box"]), "refitted child boxes"  # min/max of inputs: exact
    n = len(want["order"])
    assert np.array_equal(np.sort(got["order"]), np.arange(n, dtype=got["order"].dtype))
    assert (np.diff(got["morton"].astype(np.int64)) >= 0).all()
