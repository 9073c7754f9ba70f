"""Step counters of the occlusion kernel (a library built with -DINV_AO_STATS, see tools/build_variants.sh).

  INVERT_HIP_LIB=renderer_amd/variants/libinvert_<name>.so python tools/ao_stats.py [--closeup]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401  (one HIP runtime per process)
import objfmt
import renderer_amd as R

ap = argparse.ArgumentParser()
ap.add_argument("--closeup", action="store_true")
ap.add_argument("--rays", type=int, default=64)
args = ap.parse_args()
lib = ctypes.CDLL(R.api.LIB_PATH)
ctx = R.Context(0)
scene = R.Scene(ctx, R.Mesh.load_obj(objfmt.bunny_obj_path()))
w, h = 1920, 1080
cam = R.camera_reference(w, h)
if args.closeup:
    cam.view[12], cam.view[13], cam.view[14] = 0.0, 0.11, -0.3
prm = R.make_params(w, h, ao_rays=args.rays, flags=R.LAMBERT | R.INVERT)
out = (ctypes.c_ulonglong * 16)()
scene.render(cam, prm)
lib.inv_debug_ao_stats(out)  # clears
got = scene.render(cam, prm)
assert lib.inv_debug_ao_stats(out) == 0
steps, live, lsteps, llanes, nsteps, nlanes, pulls, fills, rays = [int(out[i]) for i in range(9)]
hits = int((got[2] != 0xFFFFFFFF).sum())
print("%s: %d hit pixels, %d rays" % (os.path.basename(R.api.LIB_PATH), hits, rays))
print("  wave steps %d (%.1f per 64 rays); live lanes per step %.1f" % (steps, steps * 64.0 / rays, live / steps))
print("  node steps %d: %.1f lanes take part (%.2f of live); leaf steps %d: %.1f lanes" % (
    nsteps, nlanes / nsteps, nlanes / nsteps / (live / steps), lsteps, llanes / max(lsteps, 1)))
print("  lane steps per ray: node %.1f, leaf %.1f; useful lane share of all steps %.3f" % (
    nlanes / rays, llanes / rays, (nlanes + llanes) / (64.0 * steps)))
print("  pulls %d, fills %d" % (pulls, fills))
