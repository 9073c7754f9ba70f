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


def pulled_back_camera(cam, distance, height=0.11):
    """The reference camera moved back along +z so that a grid of instances fits the view (same roll and lens)."""
    cam.view[12], cam.view[13], cam.view[14] = 0.0, float(np.float32(height)), float(-np.float32(distance))
    return cam
