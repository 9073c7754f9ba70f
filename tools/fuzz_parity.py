"""Random frames through the C ABI against the CPU oracle, bit for bit: scenes (the bunny, grids of copies, triangle
soups), extents (odd ones too), fields of view, eye positions, model turns, flags, kernel variants, samples per pixel,
occlusion rays and seeds that no test fixes.

    python tools/fuzz_parity.py [--frames 400] [--seed 1] [--max-width 900]

Exit status 1 at the first frame whose colour, depth or primitive ids differ (its parameters are printed).
"""
import argparse
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import common  # noqa: E402
import objfmt  # noqa: E402
import renderer_amd as R  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=400)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-width", type=int, default=900, help="widths up to this, heights up to 7/9 of it")
    a = ap.parse_args()
    O.build()
    rng = np.random.default_rng(a.seed)
    ctx = R.Context(0)
    bunny_xyzw, bunny_idx = O.load_obj(objfmt.bunny_obj_path())
    scene = None

    def new_scene():
        kind = int(rng.integers(0, 4))
        if kind <= 1:  # the bunny
            return "bunny", bunny_xyzw, bunny_idx, R.Scene(ctx, R.Mesh.from_arrays(bunny_xyzw, bunny_idx))
        if kind == 2:  # a grid of copies, flattened by the library / by numpy for the oracle
            side = int(rng.integers(2, 5))
            tr = common.grid_translations(side)
            fx, fi = common.flatten_instances(bunny_xyzw, bunny_idx, tr)
            return "grid %d" % side, fx, fi, R.Scene(ctx, R.Mesh.from_arrays(bunny_xyzw, bunny_idx), translations=tr)
        n = int(rng.integers(1, 6000))  # a soup of triangles around the place the bunny stands
        xyz, faces = objfmt.random_soup(n, seed=int(rng.integers(0, 1 << 30)), lo=(-0.08, 0.03, -0.06), hi=(0.06, 0.19, 0.06),
                                        size=float(rng.choice([0.003, 0.01, 0.04])))
        xyzw = np.zeros((len(xyz), 4), np.float32)
        xyzw[:, :3] = xyz
        fi = faces.reshape(-1).astype(np.uint32)
        return "soup %d" % n, xyzw, fi, R.Scene(ctx, R.Mesh.from_arrays(xyzw, fi))

    variants = (0, 0, 0, R.BLOCKS, R.BLOCKS, R.LANE_TRAVERSAL, R.BRUTE_FORCE)
    hits = 0
    held = []  # frames of the current scene waiting to be drawn once more, together, by inv_render_batch
    batches = 0

    def draw_held():
        # the same frames in one submission on a frame slot, into device attachments (those that can share a launch do)
        nonlocal batches
        if len(held) < 2:
            held.clear()
            return True
        bufs = [torch.empty(3, p.width * p.height, dtype=torch.int32, device="cuda") for _, p, _ in held]
        torch.cuda.synchronize()
        batch = scene.make_batch([c for c, _, _ in held], [p for _, p, _ in held],
                                 [tuple(b[k].data_ptr() for k in range(3)) for b in bufs])
        scene.render_batch(batch, slot=int(rng.integers(0, 2)))
        ctx.frames_join()
        ctx.synchronize()
        ok = True
        for (c, p, want), b in zip(held, bufs):
            g = b.cpu().numpy()
            hh, ww = p.height, p.width
            if not (np.array_equal(g[2].view(np.uint32).reshape(hh, ww), want[2])
                    and np.array_equal(g[1].view(np.uint32).reshape(hh, ww), want[1].view(np.uint32))
                    and np.array_equal(g[0].view(np.uint8).reshape(hh, ww, 4), want[0])):
                print("a frame of a batch of %d differs: %dx%d flags %d spp %d seed %d, camera bytes %s"
                      % (len(held), ww, hh, p.flags, p.spp, p.seed, bytes(c).hex()))
                ok = False
        batches += 1
        held.clear()
        return ok

    composites = 0

    def composite_equals(cam, w, h, kw, want):
        # the frame once more as several ranks would draw it, the ranks played one after another on this GPU:
        # dense (every tile dealt out, slots put back together on the host) or sparse (rank 0 draws into the frame, the
        # others' tiles arrive as a gather would deliver them and inv_deinterleave_sparse copies them in)
        nranks, tile = int(rng.integers(2, 9)), int(rng.choice([32, 64, 96, 128]))
        if rng.random() < 0.4:
            parts = [scene.render(cam, R.make_params(w, h, tile=tile, rank=r, nranks=nranks, **kw)) for r in range(nranks)]
            got = [common.slots_to_frame([p[k] for p in parts], w, h, tile, nranks) for k in range(3)]
            kind = "dense"
        else:
            kw = dict(kw, flags=kw["flags"] | R.SPARSE)
            rect = scene.hit_rect(cam, w, h)
            prm = [R.make_params(w, h, tile=tile, rank=r, nranks=nranks, **kw) for r in range(nranks)]
            _, n, _ = R.sparse_layout(prm[1], rect)
            frame = torch.full((3, h * w), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
            gathered = torch.full((nranks, 3, max(n, 1)), 0x3C3C3C3C, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            for r in range(nranks):
                o = frame if r == 0 else gathered[r]
                scene.render_device(cam, prm[r], o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
            ctx.deinterleave_sparse(prm[0], rect, gathered.data_ptr(), 3 * max(n, 1), max(n, 1), 3, frame.data_ptr(), h * w)
            ctx.synchronize()
            f = frame.cpu().numpy()
            got = [f[0].view(np.uint8).reshape(h, w, 4), f[1].view(np.float32).reshape(h, w), f[2].view(np.uint32).reshape(h, w)]
            kind = "sparse"
        ok = (np.array_equal(got[2], want[2]) and np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32))
              and np.array_equal(got[0], want[0]))
        if not ok:
            print("the %s composite of %d ranks with %d-pixel tiles differs: %dx%d %r, camera bytes %s"
                  % (kind, nranks, tile, w, h, kw, bytes(cam).hex()))
        return ok

    for i in range(a.frames):
        if i % 40 == 0:
            if scene is not None:
                if not draw_held():
                    return 1
                scene.close()
            what, xyzw, idx, scene = new_scene()
        ao_rays = int(rng.choice([4, 16, 64])) if rng.random() < 0.12 else 0
        big = min(300, a.max_width) if ao_rays else a.max_width
        w, h = int(rng.integers(1, big)), int(rng.integers(1, max(2, big * 7 // 9)))
        cam = R.camera_turntable(w, h, float(rng.uniform(0.0, 4.0))) if rng.random() < 0.5 else R.camera_reference(w, h)
        # a GLM-style perspective with another field of view (float32 arithmetic, as glm::perspective does it)
        fovy = float(rng.uniform(4.0, 140.0))
        t = np.float32(math.tan(float(np.float32(math.radians(fovy)) / np.float32(2.0))))
        aspect = np.float32(w) / np.float32(h)
        cam.proj[0] = float(np.float32(1.0) / (aspect * t))
        cam.proj[5] = float(np.float32(1.0) / t)
        # the eye somewhere in front of the model, off the axis at times
        dist = float(rng.uniform(0.25, 3.0)) * (3.0 if what.startswith("grid") else 1.0)
        cam.view[12] = float(np.float32(rng.uniform(-0.05, 0.05)))
        cam.view[13] = float(np.float32(rng.uniform(0.02, 0.2)))
        cam.view[14] = float(-np.float32(dist))
        flags = int(rng.choice([0, R.LAMBERT, R.INVERT, R.LAMBERT | R.INVERT, R.LAMBERT | R.INVERT | R.RGBA]))
        variant = int(variants[int(rng.integers(0, len(variants)))])
        spp = 1 if ao_rays else int(rng.choice([1, 1, 1, 2, 3, 4, 5, 8]))  # occlusion rays go with one sample per pixel
        seed = int(rng.integers(0, 2**31))
        prm = R.make_params(w, h, spp=spp, ao_rays=ao_rays, flags=flags | variant, seed=seed)
        got = scene.render(cam, prm)
        want = O.render_bvh(xyzw, idx, O.Camera.from_buffer_copy(bytes(cam)), O.make_params(w, h, spp, ao_rays, flags, seed))
        ok = (np.array_equal(got[2], want[2]) and np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32))
              and np.array_equal(got[0], want[0]))
        nh = int((want[2] != 0xFFFFFFFF).sum())
        hits += nh
        if not ok:
            print("frame %d (%s, %d occlusion rays) differs: %dx%d fovy %.3f dist %.3f view %r flags %d variant %d spp %d seed %d; "
                  "primid %d depth %d colour %d pixels differ"
                  % (i, what, ao_rays, w, h, fovy, dist, [cam.view[12], cam.view[13], cam.view[14]], flags, variant, spp, seed,
                     int((got[2] != want[2]).sum()), int((got[1].view(np.uint32) != want[1].view(np.uint32)).sum()),
                     int((got[0] != want[0]).any(axis=-1).sum())))
            ys, xs = np.nonzero((got[2] != want[2]) | (got[1].view(np.uint32) != want[1].view(np.uint32)) | (got[0] != want[0]).any(axis=-1))
            for y, x in list(zip(ys, xs))[:4]:
                print("  pixel (%d, %d): primid %d / oracle %d, depth %r / %r, colour %r / %r"
                      % (x, y, got[2][y, x], want[2][y, x], float(got[1][y, x]), float(want[1][y, x]), got[0][y, x].tolist(), want[0][y, x].tolist()))
            for name, v in (("tiles", 0), ("blocks", R.BLOCKS), ("lanes", R.LANE_TRAVERSAL), ("brute force", R.BRUTE_FORCE)):
                g = scene.render(cam, R.make_params(w, h, spp=spp, ao_rays=ao_rays, flags=flags | v, seed=seed))
                same = (np.array_equal(g[2], want[2]) and np.array_equal(g[1].view(np.uint32), want[1].view(np.uint32))
                        and np.array_equal(g[0], want[0]))
                print("  %-12s %s" % (name, "equals the oracle" if same else "differs"))
            print("  camera bytes:", bytes(cam).hex())
            return 1
        if rng.random() < 0.2:
            composites += 1
            if not composite_equals(cam, w, h, dict(spp=spp, ao_rays=ao_rays, flags=flags | variant, seed=seed), want):
                return 1
        if ao_rays == 0 and w * h < 3000000:
            held.append((cam, prm, want))
            if len(held) == int(rng.integers(2, 10)) or len(held) >= 9:
                if not draw_held():
                    return 1
        if (i + 1) % 50 == 0:
            print("%d frames bit-exact so far (%d hit pixels)" % (i + 1, hits), flush=True)
    if not draw_held():
        return 1
    print("%d random frames bit-exact against the oracle (%d hit pixels in all); %d batches of them drawn again through "
          "inv_render_batch and %d as the shards of several ranks put together again, bit-exact too"
          % (a.frames, hits, batches, composites))
    return 0


if __name__ == "__main__":
    sys.exit(main())
