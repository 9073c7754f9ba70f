"""How far a Vulkan rasteriser's fixed-function rules can move pixels away from the ray-cast goldens.

    python tests/golden/make_raster_bounds.py      (CPU, numpy; rewrites the "vulkan_raster_bounds" entry of pins.json)

The reference draws with vkCmdDrawIndexed (src/invert.cpp:828; raster state :1331-1337, one sample :1375-1378): which
pixels a triangle covers is then decided by the implementation's rasteriser, with the top-left fill rule and vertex
positions snapped to a fixed-point grid (subPixelPrecisionBits).  The goldens come from ray casting through pixel
centres with inclusive edges and unsnapped float32 positions.  For configs 1 and 2 this script rasterises the bunny
with tests/raster_np.py under each rule and records, per variant, how many pixels get another primitive id than
the golden (and which), how many change between hit and miss, and the largest depth difference where the ids agree.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import common  # noqa: E402
import raster_np  # noqa: E402
from test_oracle_pins import _bunny_from_the_decimal_fields  # noqa: E402

VARIANTS = {
    "inclusive_edges_unsnapped": dict(),
    "top_left_unsnapped": dict(top_left=True),
    "top_left_8_subpixel_bits": dict(top_left=True, subpixel_bits=8),
    "top_left_4_subpixel_bits": dict(top_left=True, subpixel_bits=4),
}
CONFIGS = [("512x512", "bunny_512"), ("1920x1080", "bunny_1080_rect")]


def bounds(size, golden, variant):
    w, h = [int(v) for v in size.split("x")]
    xyz, faces = _bunny_from_the_decimal_fields()
    cam = common.pins()["cameras"][size]
    primid, depth, _, _, nfront = raster_np.rasterise(xyz, faces, common.unhex(cam["model"]), common.unhex(cam["view"]),
                                                      common.unhex(cam["proj"]), w, h, **VARIANTS[variant])
    _, gd, gp = common.golden_frame(golden)
    differ = gp != primid
    miss = np.uint32(0xFFFFFFFF)
    same = ~differ & (gp != miss)
    ys, xs = np.nonzero(differ)
    return {
        "covered": int((primid != miss).sum()), "golden_covered": int((gp != miss).sum()), "front_faces": int(nfront),
        "primitive_id_differs": int(differ.sum()),
        "hit_in_golden_only": int((differ & (primid == miss)).sum()), "hit_in_variant_only": int((differ & (gp == miss)).sum()),
        "another_triangle": int((differ & (primid != miss) & (gp != miss)).sum()),
        "max_depth_difference_in_2^-24": float(np.abs(gd[same].astype(np.float64) - depth[same]).max() / 2.0 ** -24),
        # (y, x, golden id, variant id), at most 64 of them
        "pixels": [[int(y), int(x), int(gp[y, x]), int(primid[y, x])] for y, x in list(zip(ys, xs))[:64]],
    }


def main():
    out = {size: {v: bounds(size, golden, v) for v in VARIANTS} for size, golden in CONFIGS}
    path = os.path.join(HERE, "pins.json")
    with open(path) as f:
        pins = json.load(f)
    pins["vulkan_raster_bounds"] = out
    with open(path, "w") as f:
        json.dump(pins, f, indent=1)
        f.write("\n")
    for size in out:
        for v, r in out[size].items():
            print(size, v, {k: r[k] for k in r if k != "pixels"})


if __name__ == "__main__":
    main()
