"""inv_render_noalloc (the entry point that allocates nothing) against inv_render, through the C ABI.

Bar: bit-exact colour bytes, depth bits and primitive ids.  Two calls in a row share the GPU's static occlusion
queue, the second with a smaller input, so stale contents of the first call would show.
"""
import numpy as np
import pytest
import torch

import renderer_amd as R

pytestmark = pytest.mark.gpu


def _both(scene, ctx, cam, prm):
    """(planes drawn by inv_render, planes drawn by inv_render_noalloc), each [3, h*w] int32 on the host."""
    n = prm.width * prm.height
    a = torch.full((3, n), 0x55, dtype=torch.int32, device="cuda")
    b = torch.full((3, n), 0x66, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    scene.render_device(cam, prm, a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr())
    scene.render_noalloc(cam, prm, b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr())
    ctx.synchronize()
    return a.cpu().numpy(), b.cpu().numpy()


def test_noalloc_entry_point_matches_inv_render(bunny_path):
    ctx = R.Context(0)
    scene = R.Scene(ctx, R.Mesh.load_obj(bunny_path))
    try:
        # first call: 1080p, 64 occlusion rays per hit pixel (25 139 queued pixels)
        w, h = 1920, 1080
        prm = R.make_params(w, h, ao_rays=64, flags=R.LAMBERT | R.INVERT)
        want, got = _both(scene, ctx, R.camera_reference(w, h), prm)
        assert np.array_equal(got, want)
        assert int((got[2] != -1).sum()) == 25139
        # second call, smaller input: the queue still holds the first call's records and counts
        w, h = 640, 360
        prm = R.make_params(w, h, ao_rays=16, seed=3, flags=R.LAMBERT)
        want, got = _both(scene, ctx, R.camera_reference(w, h), prm)
        assert np.array_equal(got, want)
        assert 2000 < int((got[2] != -1).sum()) < 25139
        # no occlusion rays: the queue is not touched at all
        prm = R.make_params(w, h, spp=2, seed=9, flags=R.INVERT)
        want, got = _both(scene, ctx, R.camera_reference(w, h), prm)
        assert np.array_equal(got, want)
    finally:
        scene.close()
        ctx.close()


def test_noalloc_entry_point_leaves_free_memory_as_it_is(bunny_path):
    """A scene that inv_render has never drawn (no occlusion queue, no wide node records, no bins exist yet): the
    first-ever occlusion draw through inv_render_noalloc alone leaves the driver's count of free device bytes
    unchanged, and its frame equals the one a second scene gets from inv_render."""
    ctx = R.Context(0)
    fresh = R.Scene(ctx, R.Mesh.load_obj(bunny_path))
    other = R.Scene(ctx, R.Mesh.load_obj(bunny_path))
    try:
        w, h = 1280, 720
        cam, prm = R.camera_reference(w, h), R.make_params(w, h, ao_rays=32, seed=5, flags=R.LAMBERT | R.INVERT)
        a = torch.full((3, w * h), 0x55, dtype=torch.int32, device="cuda")
        b = torch.full((3, w * h), 0x66, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]  # hipMemGetInfo: the driver's figure, not torch's allocator
        for _ in range(2):
            fresh.render_noalloc(cam, prm, b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr())
        ctx.synchronize()
        assert torch.cuda.mem_get_info()[0] == free_before
        other.render_device(cam, prm, a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr())
        ctx.synchronize()
        assert torch.cuda.mem_get_info()[0] < free_before  # the ordinary draw does allocate (queue, wide records)
        assert np.array_equal(b.cpu().numpy(), a.cpu().numpy())
    finally:
        fresh.close()
        other.close()
        ctx.close()


def test_noalloc_entry_point_rejects_what_needs_memory(bunny_path):
    ctx = R.Context(0)
    scene = R.Scene(ctx, R.Mesh.load_obj(bunny_path))
    try:
        w, h = 1920, 1080
        buf = torch.full((3, w * h), 0x77, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        # from close by the bunny fills the frame: more pixels can be hit than the static queue holds
        cam = R.camera_reference(w, h)
        cam.view[12], cam.view[13], cam.view[14] = 0.0, 0.11, -0.3
        x0, y0, x1, y1 = scene.hit_rect(cam, w, h)
        assert (x1 - x0) * (y1 - y0) > 262144
        with pytest.raises(R.InvertError) as e:
            scene.render_noalloc(cam, R.make_params(w, h, ao_rays=64), buf[0].data_ptr(), buf[1].data_ptr(), buf[2].data_ptr())
        assert e.value.code == R.api.ERR_ARG and "262144" in str(e.value)
        ctx.synchronize()
        assert bool((buf == 0x77).all())  # nothing was drawn
        # the same frame without occlusion rays needs no queue and is drawn
        scene.render_noalloc(cam, R.make_params(w, h), buf[0].data_ptr(), buf[1].data_ptr(), buf[2].data_ptr())
        ctx.synchronize()
        assert int((buf[2] != -1).sum()) > 262144
        # host attachments need staging memory
        import ctypes as C
        host = np.zeros(w * h, np.uint32)
        out = R.api.Outputs(None, None, host.ctypes.data, 0)
        prm = R.make_params(w, h)
        rc = R.api.lib().inv_render_noalloc(scene._h, C.byref(cam), C.byref(prm), C.byref(out))
        assert rc == R.api.ERR_ARG
    finally:
        scene.close()
        ctx.close()
