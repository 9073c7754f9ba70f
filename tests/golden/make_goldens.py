"""Freeze the oracle's outputs for the plumbing configuration as committed fixtures.

    python tests/golden/make_goldens.py

Writes
  tests/golden/pins.json          -- loader / camera facts as hex floats and digests
  tests/golden/bunny_512.npz      -- primitive id, depth and colour of the 512x512 reference-camera
                                     render, cropped to the bunny's bounding rectangle
  tests/golden/bunny_1080_rect.npz-- the same at 1920x1080 (parity mode), cropped

The reference ships no golden of its own (parity unpinned); these pin this repo's
oracle so later changes to it are visible.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import objfmt  # noqa: E402
from oracle import oracle as O  # noqa: E402


def hexf(a):
    return [float(v).hex() for v in np.asarray(a, np.float32).reshape(-1)]


def crop(img, rect):
    x0, y0, x1, y1 = rect
    return np.ascontiguousarray(img[y0:y1, x0:x1])


def main():
    xyzw, idx = O.load_obj(objfmt.bunny_obj_path())
    pins = {
        "obj_sha256": objfmt.bunny_sha256(),
        "nv": int(xyzw.shape[0]),
        "nidx": int(idx.size),
        "vertices_sha256": hashlib.sha256(np.ascontiguousarray(xyzw[:, :3]).tobytes()).hexdigest(),
        "indices_sha256": hashlib.sha256(idx.tobytes()).hexdigest(),
        "first_vertices": [hexf(xyzw[i, :3]) for i in range(3)],
        "bbox_min": hexf(xyzw[:, :3].min(0)),
        "bbox_max": hexf(xyzw[:, :3].max(0)),
        "cameras": {},
        "frames": {},
    }
    for (w, h) in [(512, 512), (1920, 1080), (3840, 2160)]:
        cam = O.camera_reference(w, h)
        pins["cameras"]["%dx%d" % (w, h)] = {
            "model": hexf(cam.model), "view": hexf(cam.view), "proj": hexf(cam.proj), "cam_angle": hexf(cam.cam_angle)}
        pins["frames"]["%dx%d" % (w, h)] = hexf(O.frame_setup(xyzw, cam))
    for name, (w, h) in (("bunny_512", (512, 512)), ("bunny_1080_rect", (1920, 1080))):
        cam = O.camera_reference(w, h)
        prm = O.make_params(w, h)
        _, pr, _ = O.raster(xyzw, idx, cam, w, h)
        ys, xs = np.nonzero(pr != 0xFFFFFFFF)
        rect = (max(int(xs.min()) - 8, 0), max(int(ys.min()) - 8, 0), min(int(xs.max()) + 9, w), min(int(ys.max()) + 9, h))
        color, depth, primid = O.render_bruteforce(xyzw, idx, cam, prm, rect=rect)
        hit = primid != 0xFFFFFFFF
        # nothing outside the crop is hit: check with the BVH oracle over the whole frame
        _, _, p_full = O.render_bvh(xyzw, idx, cam, prm)
        assert np.array_equal(p_full, primid), "hits outside the crop rectangle"
        np.savez_compressed(os.path.join(HERE, name + ".npz"), rect=np.array(rect, np.int32),
                            size=np.array([w, h], np.int32), color=crop(color, rect), depth=crop(depth, rect),
                            primid=crop(primid, rect))
        pins[name] = {"hits": int(hit.sum()), "rect": list(rect),
                      "depth_min": float(depth[hit].min()).hex(), "depth_max": float(depth[hit].max()).hex(),
                      "primid_sha256": hashlib.sha256(primid.tobytes()).hexdigest(),
                      "depth_sha256": hashlib.sha256(depth.tobytes()).hexdigest(),
                      "color_sha256": hashlib.sha256(color.tobytes()).hexdigest()}
        print(name, pins[name])
    with open(os.path.join(HERE, "pins.json"), "w") as f:
        json.dump(pins, f, indent=1)
    print("wrote pins.json")


if __name__ == "__main__":
    main()
