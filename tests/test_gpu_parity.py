"""GPU parity tests: the HIP path through the C ABI against the CPU oracle and the committed goldens.

Bar: bit-exact primitive id, depth bits and colour bytes (integer / byte work, no tolerance).
"""
import numpy as np
import pytest

import common
import objfmt
import renderer_amd as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bunny_scene(gpu_ctx, bunny_path):
    mesh = R.Mesh.load_obj(bunny_path)
    scene = R.Scene(gpu_ctx, mesh)
    yield scene
    scene.close()


_oracle_params = common.oracle_params


def test_vertex_stage_matches_oracle(oracle, bunny_scene, bunny_mesh):
    cam = R.camera_reference(1920, 1080)
    clip, col = bunny_scene.vertex_shader(cam)
    oclip, ocol = oracle.vertex_shader(bunny_mesh[0], oracle.camera_reference(1920, 1080))
    assert np.array_equal(clip.view(np.uint32), oclip.view(np.uint32))
    assert np.array_equal(col.view(np.uint32), ocol.view(np.uint32))


MODES = [0, R.BRUTE_FORCE, R.LANE_TRAVERSAL, R.BLOCKS]  # tile kernel (8x8 tiles: default), brute force, per-lane
# stack traversal, tile kernel with 16x16 blocks (what inv_render_batch uses)


@pytest.mark.parametrize("flags", MODES)
def test_config1_512_against_golden(bunny_scene, flags):
    got = bunny_scene.render(R.camera_reference(512, 512), R.make_params(512, 512, flags=flags))
    common.assert_frames_equal(got, common.golden_frame("bunny_512"), "config 1")
    assert int((got[2] != 0xFFFFFFFF).sum()) == 5651


@pytest.mark.parametrize("mode", [0, R.LANE_TRAVERSAL, R.BLOCKS])
def test_config2_1080_against_golden(bunny_scene, mode):
    got = bunny_scene.render(R.camera_reference(1920, 1080), R.make_params(1920, 1080, flags=mode))
    common.assert_frames_equal(got, common.golden_frame("bunny_1080_rect"), "config 2 (parity mode)")
    assert int((got[2] != 0xFFFFFFFF).sum()) == 25139


@pytest.mark.parametrize("flags", [R.LAMBERT, R.INVERT, R.LAMBERT | R.INVERT, R.LAMBERT | R.INVERT | R.RGBA])
@pytest.mark.parametrize("brute", MODES)
def test_shading_modes_against_oracle(oracle, bunny_scene, bunny_mesh, flags, brute):
    w, h = 640, 360
    prm = R.make_params(w, h, flags=flags | brute)
    got = bunny_scene.render(R.camera_reference(w, h), prm)
    want = oracle.render_bvh(*bunny_mesh, oracle.camera_reference(w, h), _oracle_params(oracle, prm))
    common.assert_frames_equal(got, want, "flags=%d" % flags)


def test_config2_timed_mode_1080_against_oracle(oracle, bunny_scene, bunny_mesh):
    """1080p primary + Lambert + invert: the workload bench.py times."""
    prm = R.make_params(1920, 1080, flags=R.LAMBERT | R.INVERT)
    got = bunny_scene.render(R.camera_reference(1920, 1080), prm)
    want = oracle.render_bvh(*bunny_mesh, oracle.camera_reference(1920, 1080), _oracle_params(oracle, prm))
    common.assert_frames_equal(got, want, "config 2 (Lambert + invert)")


@pytest.mark.parametrize("brute", MODES)
def test_jittered_samples_against_oracle(oracle, bunny_scene, bunny_mesh, brute):
    w, h = 320, 240
    prm = R.make_params(w, h, spp=4, seed=11, flags=R.LAMBERT | brute)
    got = bunny_scene.render(R.camera_reference(w, h), prm)
    want = oracle.render_bvh(*bunny_mesh, oracle.camera_reference(w, h), _oracle_params(oracle, prm))
    common.assert_frames_equal(got, want, "spp=4")


@pytest.mark.parametrize("spp", [2, 4, 5, 8])
def test_samples_in_the_sparse_scene_tile_kernel(oracle, bunny_scene, bunny_mesh, spp):
    """720p keeps the bunny's triangles larger than a pixel (the variant of the tile kernel that holds a tile's
    candidates at once): below four samples the waves share each sample's records, from four on each wave carries
    whole samples (5: a last round with one wave busy)."""
    w, h = 1280, 720
    prm = R.make_params(w, h, spp=spp, seed=5, flags=R.LAMBERT | R.INVERT)
    got = bunny_scene.render(R.camera_reference(w, h), prm)
    want = oracle.render_bvh(*bunny_mesh, oracle.camera_reference(w, h), _oracle_params(oracle, prm))
    common.assert_frames_equal(got, want, "spp=%d" % spp)


@pytest.mark.parametrize("size", [(1, 1), (7, 5), (64, 64), (129, 65), (1000, 3)])
def test_odd_extents(oracle, bunny_scene, bunny_mesh, size):
    w, h = size
    prm = R.make_params(w, h)
    got = bunny_scene.render(R.camera_reference(w, h), prm)
    want = oracle.render_bvh(*bunny_mesh, oracle.camera_reference(w, h), _oracle_params(oracle, prm))
    common.assert_frames_equal(got, want, "%dx%d" % size)


def test_moved_camera(oracle, bunny_scene, bunny_mesh):
    """A camera other than the reference's: closer, off-axis, rotated model matrix."""
    w, h = 400, 300
    cam = R.camera_reference(w, h)
    view = np.array(cam.view, np.float32).reshape(4, 4)
    view[3, :3] = [0.03, 0.11, -0.35]  # translation column
    c, s = np.float32(np.cos(0.5)), np.float32(np.sin(0.5))
    model = np.eye(4, dtype=np.float32)
    model[0, 0], model[0, 2], model[2, 0], model[2, 2] = c, -s, s, c
    cam.view[:] = view.reshape(-1).tolist()
    cam.model[:] = model.reshape(-1).tolist()
    ocam = oracle.Camera.from_buffer_copy(bytes(cam))
    for flags in (0, R.LAMBERT, R.LANE_TRAVERSAL, R.BLOCKS | R.LAMBERT):
        prm = R.make_params(w, h, flags=flags)
        got = bunny_scene.render(cam, prm)
        want = oracle.render_bvh(*bunny_mesh, ocam, _oracle_params(oracle, prm))
        common.assert_frames_equal(got, want, "moved camera")
        assert (got[2] != 0xFFFFFFFF).sum() > 5000


def test_random_soup_and_degenerate_triangles(oracle, gpu_ctx):
    """Unsorted overlapping triangles (depth test and tie-break matter), plus zero-area and repeated faces."""
    xyz, faces = objfmt.random_soup(600, seed=5)
    faces = np.concatenate([faces, faces[:50], np.array([[0, 0, 1], [2, 2, 2]], np.uint32)])  # duplicates: lower id wins
    text = objfmt.from_floats(xyz, faces).text()
    mesh = R.Mesh.parse_obj(text)
    scene = R.Scene(gpu_ctx, mesh)
    oxyzw, oidx = oracle.parse_obj(text)
    w, h = 300, 300
    for flags in (0, R.BRUTE_FORCE, R.LAMBERT, R.LANE_TRAVERSAL, R.BLOCKS):
        prm = R.make_params(w, h, flags=flags)
        got = scene.render(R.camera_reference(w, h), prm)
        want = oracle.render_bruteforce(oxyzw, oidx, oracle.camera_reference(w, h), _oracle_params(oracle, prm))
        common.assert_frames_equal(got, want, "soup flags=%d" % flags)
    assert (got[2] != 0xFFFFFFFF).sum() > 2000
    scene.close()


def test_tiny_scenes(oracle, gpu_ctx):
    """0, 1 and 2 triangles (no LBVH below 2)."""
    tri = np.array([[-0.05, 0.05, 0.0], [0.05, 0.05, 0.0], [0.0, 0.15, 0.02],
                    [-0.02, 0.08, 0.03], [0.0, 0.12, 0.03], [0.02, 0.08, 0.03]])
    for nf in (0, 1, 2):
        faces = np.array([[0, 1, 2], [3, 5, 4]], np.uint32)[:nf]
        text = objfmt.from_floats(tri, faces).text()
        scene = R.Scene(gpu_ctx, R.Mesh.parse_obj(text))
        oxyzw, oidx = oracle.parse_obj(text)
        prm = R.make_params(128, 128)
        got = scene.render(R.camera_reference(128, 128), prm)
        want = oracle.render_bruteforce(oxyzw, oidx, oracle.camera_reference(128, 128), _oracle_params(oracle, prm))
        common.assert_frames_equal(got, want, "%d triangles" % nf)
        scene.close()


@pytest.mark.parametrize("nranks,tile", [(2, 64), (3, 32), (8, 64)])
def test_tile_sharding_and_deinterleave(bunny_scene, gpu_ctx, nranks, tile):
    """Every rank's tile slots, put back together, equal the single-rank frame (no data-path exchange)."""
    import torch
    w, h = 1920, 1080
    cam = R.camera_reference(w, h)
    flags = R.LAMBERT | R.INVERT
    full = bunny_scene.render(cam, R.make_params(w, h, flags=flags))
    parts = [bunny_scene.render(cam, R.make_params(w, h, flags=flags, tile=tile, rank=r, nranks=nranks))
             for r in range(nranks)]
    for k in range(3):
        frame = common.slots_to_frame([p[k] for p in parts], w, h, tile, nranks)
        assert np.array_equal(frame, full[k])
    # device de-interleave kernel (what rank 0 runs after the gather)
    gathered = torch.from_numpy(np.concatenate([p[2].reshape(-1) for p in parts]).view(np.int32)).cuda()
    out = torch.empty(h * w, dtype=torch.int32, device="cuda")
    gpu_ctx.deinterleave(R.make_params(w, h, tile=tile, nranks=nranks), gathered.data_ptr(), out.data_ptr())
    gpu_ctx.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint32).reshape(h, w), full[2])


@pytest.mark.parametrize("nranks,tile", [(3, 64), (5, 32)])
def test_dense_deinterleave_with_an_idle_root_and_the_default_stride(bunny_scene, gpu_ctx, nranks, tile):
    """INV_ROOT_IDLE, dense layout, rank_stride left at 0 ("back to back"): the buffers of the ranks lie one drawing
    rank's slot count apart, rank 0's (unused) part included."""
    import torch
    w, h = 1000, 600
    cam = R.camera_reference(w, h)
    full = bunny_scene.render(cam, R.make_params(w, h))
    prm = [R.make_params(w, h, tile=tile, rank=r, nranks=nranks, flags=R.ROOT_IDLE) for r in range(nranks)]
    npix, _ = R.tile_layout(prm[1])
    assert R.tile_layout(prm[0])[0] == 0 and npix > 0
    gathered = torch.full((nranks, npix), 0x3C3C3C3C, dtype=torch.int32, device="cuda")
    scratch = torch.empty(2, npix, dtype=torch.int32, device="cuda")
    for r in range(1, nranks):
        bunny_scene.render_device(cam, prm[r], scratch[0].data_ptr(), scratch[1].data_ptr(), gathered[r].data_ptr())
    out = torch.empty(h * w, dtype=torch.int32, device="cuda")
    gpu_ctx.deinterleave(prm[0], gathered.data_ptr(), out.data_ptr())  # rank_stride = 0
    gpu_ctx.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint32).reshape(h, w), full[2])


_sparse_composite_single_gpu = common.sparse_composite_single_gpu


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("nranks,tile,w,h", [(2, 64, 1920, 1080), (8, 64, 1920, 1080), (3, 32, 500, 301), (4, 64, 333, 250)])
def test_sparse_composite_equals_single_rank_frame(bunny_scene, gpu_ctx, mode, nranks, tile, w, h):
    """Only the tiles under the projected scene bounds are dealt out; the composite is the single-rank frame."""
    cam = R.camera_reference(w, h)
    flags = R.LAMBERT | R.INVERT | mode
    full = bunny_scene.render(cam, R.make_params(w, h, flags=flags))
    got, rect, grid, n = _sparse_composite_single_gpu(bunny_scene, gpu_ctx, cam, w, h, nranks, tile, flags=flags)
    common.assert_frames_equal(got, full, "sparse composite")
    # the dealt grid covers every hit pixel and is much smaller than the frame
    ys, xs = np.nonzero(full[2] != 0xFFFFFFFF)
    assert rect[0] <= xs.min() and xs.max() < rect[2] and rect[1] <= ys.min() and ys.max() < rect[3]
    assert grid[2] * grid[3] * tile * tile < 0.2 * w * h
    assert n == -(-grid[2] * grid[3] // nranks) * tile * tile


def test_batched_sparse_composite(bunny_scene, gpu_ctx):
    """Four frames per submission and rank (inv_render_batch on shards of a sparse composite), the ranks played one
    after another on this GPU, one inv_deinterleave_sparse over the 12 planes: every frame equals the single-rank
    frame."""
    import torch
    w, h, nranks, tile, F = 1280, 720, 3, 32, 4
    cam = R.camera_reference(w, h)
    flags = R.LAMBERT | R.INVERT | R.SPARSE
    full = bunny_scene.render(cam, R.make_params(w, h, flags=flags & ~R.SPARSE))
    rect = bunny_scene.hit_rect(cam, w, h)
    prm = [R.make_params(w, h, flags=flags, tile=tile, rank=r, nranks=nranks) for r in range(nranks)]
    _, n, _ = R.sparse_layout(prm[1], rect)
    frames = torch.full((F, 3, h * w), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    gathered = torch.full((nranks, F, 3, n), 0x3C3C3C3C, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for r in range(nranks):
        outs = [frames[j] if r == 0 else gathered[r, j] for j in range(F)]
        bunny_scene.render_batch(bunny_scene.make_batch([cam] * F, [prm[r]] * F,
                                                        [(o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr()) for o in outs]))
    gpu_ctx.deinterleave_sparse(prm[0], rect, gathered.data_ptr(), F * 3 * n, n, 3 * F, frames.data_ptr(), h * w)
    gpu_ctx.synchronize()
    host = frames.cpu().numpy()
    for j in range(F):
        got = (host[j, 0].view(np.uint8).reshape(h, w, 4), host[j, 1].view(np.float32).reshape(h, w),
               host[j, 2].view(np.uint32).reshape(h, w))
        common.assert_frames_equal(got, full, "batched sparse composite, frame %d" % j)


def test_sparse_composite_with_jitter_and_occlusion(bunny_scene, gpu_ctx):
    w, h = 640, 360
    cam = R.camera_reference(w, h)
    for kw in (dict(spp=4, seed=7), dict(ao_rays=16, seed=3)):
        flags = R.LAMBERT | R.INVERT
        full = bunny_scene.render(cam, R.make_params(w, h, flags=flags, **kw))
        got, _, _, _ = _sparse_composite_single_gpu(bunny_scene, gpu_ctx, cam, w, h, 3, 32, flags=flags, **kw)
        common.assert_frames_equal(got, full, "sparse composite %r" % (kw,))


def test_sparse_composite_with_nothing_on_screen(bunny_scene, gpu_ctx):
    """Scene behind the camera: no tile is dealt out, rank 0 clears the whole frame, nothing is gathered."""
    w, h = 256, 192
    cam = R.camera_reference(w, h)
    cam.view[14] = 5.0  # move the scene behind the eye
    full = bunny_scene.render(cam, R.make_params(w, h))
    assert (full[2] == 0xFFFFFFFF).all()
    got, rect, grid, n = _sparse_composite_single_gpu(bunny_scene, gpu_ctx, cam, w, h, 4, 64)
    assert n == 0 and grid[2] * grid[3] == 0
    common.assert_frames_equal(got, full, "empty sparse composite")


@pytest.mark.parametrize("mode", MODES)
def test_turntable_frames_reuse_the_scene(oracle, bunny_scene, bunny_mesh, mode):
    """A sequence of frames with the model turning and the extent changing, all from one scene build: rays are cast
    in object space, so neither the triangles nor the LBVH depend on the camera."""
    for t, (w, h) in ((0.3, (1920, 1080)), (1.0, (640, 360)), (2.5, (1280, 720)), (3.75, (1920, 1080))):
        prm = R.make_params(w, h, flags=R.LAMBERT | R.INVERT | mode)
        got = bunny_scene.render(R.camera_turntable(w, h, t), prm)
        want = oracle.render_bvh(*bunny_mesh, oracle.camera_turntable(w, h, t), _oracle_params(oracle, prm))
        common.assert_frames_equal(got, want, "t=%.2f %dx%d" % (t, w, h))
        assert (got[2] != 0xFFFFFFFF).sum() > 2000


def test_device_outputs_on_torch_stream(bunny_scene, bunny_path):
    """Device-pointer outputs on the caller's stream (the path bench.py times)."""
    import torch
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx = R.Context(0, stream=stream.cuda_stream)
    scene = R.Scene(ctx, R.Mesh.load_obj(bunny_path))
    w, h = 1920, 1080
    color = torch.empty(h, w, 4, dtype=torch.uint8, device="cuda")
    depth = torch.empty(h, w, dtype=torch.float32, device="cuda")
    primid = torch.empty(h, w, dtype=torch.int32, device="cuda")
    scene.render_device(R.camera_reference(w, h), R.make_params(w, h), color.data_ptr(), depth.data_ptr(), primid.data_ptr())
    stream.synchronize()  # only the caller's stream: the draw must have been enqueued on it
    torch.cuda.set_stream(torch.cuda.default_stream())
    got = (color.cpu().numpy(), depth.cpu().numpy(), primid.cpu().numpy().view(np.uint32))
    common.assert_frames_equal(got, common.golden_frame("bunny_1080_rect"), "device outputs")
    scene.close()
    ctx.close()


def test_frames_in_flight_match_the_oracle(oracle, bunny_path, bunny_mesh):
    """Four different draws (turning model, different extents, jitter, occlusion rays) enqueued on the four frame
    slots before anything is waited for, each into attachments of its own; then the same slots again with other
    frames (a slot's second draw queues behind its first)."""
    import torch
    torch.cuda.set_device(0)
    ctx = R.Context(0)
    scene = R.Scene(ctx, R.Mesh.load_obj(bunny_path))
    jobs = [
        (0.3, 1920, 1080, dict(flags=R.LAMBERT | R.INVERT)),
        (1.0, 640, 360, dict(flags=R.LAMBERT, spp=4, seed=3)),
        (2.5, 1280, 720, dict(flags=R.LAMBERT | R.INVERT, ao_rays=16)),
        (3.75, 800, 600, dict(flags=R.LANE_TRAVERSAL)),
        (4.1, 1024, 512, dict(flags=R.INVERT | R.RGBA)),
        (0.0, 1920, 1080, dict(flags=0)),
        (5.0, 320, 240, dict(flags=R.LAMBERT, ao_rays=5)),
        (6.2, 512, 512, dict(flags=R.BRUTE_FORCE)),
    ]
    bufs = [torch.zeros(3, h * w, dtype=torch.int32, device="cuda") for (t, w, h, kw) in jobs]
    torch.cuda.synchronize()  # the fills run on torch's stream, which the slots' streams do not wait for
    for i, (t, w, h, kw) in enumerate(jobs):
        planes = bufs[i]
        scene.render_in_flight(i % R.api.MAX_FRAMES_IN_FLIGHT, R.camera_turntable(w, h, t), R.make_params(w, h, **kw),
                               planes[0].data_ptr(), planes[1].data_ptr(), planes[2].data_ptr())
    for slot in range(R.api.MAX_FRAMES_IN_FLIGHT):
        ctx.frame_wait(slot)  # only the slots' own streams are waited for
    for planes, (t, w, h, kw) in zip(bufs, jobs):
        host = planes.cpu().numpy()
        got = (host[0].view(np.uint8).reshape(h, w, 4), host[1].view(np.float32).reshape(h, w),
               host[2].view(np.uint32).reshape(h, w))
        prm = R.make_params(w, h, **kw)
        want = oracle.render_bvh(*bunny_mesh, oracle.camera_turntable(w, h, t), _oracle_params(oracle, prm))
        common.assert_frames_equal(got, want, "in flight t=%.2f %dx%d" % (t, w, h))
    # host-pointer outputs on a slot: the call itself waits for that slot
    w, h = 333, 250
    prm = R.make_params(w, h, flags=R.LAMBERT | R.INVERT)
    got = scene.render_in_flight_host(2, R.camera_reference(w, h), prm)
    want = oracle.render_bvh(*bunny_mesh, oracle.camera_reference(w, h), _oracle_params(oracle, prm))
    common.assert_frames_equal(got, want, "in flight, host outputs")
    with pytest.raises(R.InvertError):
        scene.render_in_flight(R.api.MAX_FRAMES_IN_FLIGHT, R.camera_reference(w, h), prm, bufs[0][0].data_ptr())
    with pytest.raises(R.InvertError):
        ctx.frame_wait(R.api.MAX_FRAMES_IN_FLIGHT)
    scene.close()
    ctx.close()


def test_batched_draws_match_the_oracle(oracle, bunny_path, bunny_mesh):
    """inv_render_batch: frames of different cameras, extents and flags in one submission, on frame slots and on the
    context's stream; frames that can share launches of the tile kernel do, the others are drawn in between."""
    import torch
    torch.cuda.set_device(0)
    ctx = R.Context(0)
    scene = R.Scene(ctx, R.Mesh.load_obj(bunny_path))
    # Ten frames that can share launches (at least 720p: few triangles per tile; fewer than four samples per pixel; no
    # occlusion rays): one launch of eight and one of two, with frames in between that are drawn on their own (tiny
    # extents, where triangles are smaller than pixels; five samples per pixel).
    shared = [(0.1 * i, w, h, kw) for i, (w, h, kw) in enumerate([
        (1920, 1080, dict(flags=R.LAMBERT | R.INVERT)), (1280, 720, dict(flags=0)), (64, 64, dict(flags=R.LAMBERT)),
        (1920, 1080, dict(flags=0)), (1600, 900, dict(flags=R.LAMBERT, spp=2, seed=9)), (7, 5, dict(flags=0)),
        (1366, 768, dict(flags=R.INVERT | R.RGBA)), (2560, 1440, dict(flags=R.LAMBERT)), (1001, 3, dict(flags=0)),
        (1920, 1200, dict(flags=R.LAMBERT, spp=3, seed=3)), (1280, 1024, dict(flags=R.INVERT)),
        (640, 360, dict(flags=R.LAMBERT, spp=5, seed=4)), (3840, 2160, dict(flags=R.LAMBERT | R.INVERT)),
        (1440, 900, dict(flags=0)),
    ])]
    mixed = [(0.7, 640, 360, dict(flags=R.LAMBERT, ao_rays=8)), (1.9, 800, 600, dict(flags=R.LANE_TRAVERSAL)),
             (2.2, 512, 512, dict(flags=R.LAMBERT | R.INVERT)), (2.9, 1280, 720, dict(flags=R.LAMBERT))]
    for jobs, slot in ((shared, 1), (mixed, 2), (shared[:3], R.api.CONTEXT_STREAM)):
        bufs = [torch.zeros(3, h * w, dtype=torch.int32, device="cuda") for (t, w, h, kw) in jobs]
        torch.cuda.synchronize()
        batch = scene.make_batch([R.camera_turntable(w, h, t) for (t, w, h, kw) in jobs],
                                 [R.make_params(w, h, **kw) for (t, w, h, kw) in jobs],
                                 [(b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr()) for b in bufs])
        scene.render_batch(batch, slot)
        if slot == R.api.CONTEXT_STREAM:
            ctx.synchronize()
        else:
            ctx.frame_wait(slot)
        for planes, (t, w, h, kw) in zip(bufs, jobs):
            host = planes.cpu().numpy()
            got = (host[0].view(np.uint8).reshape(h, w, 4), host[1].view(np.float32).reshape(h, w),
                   host[2].view(np.uint32).reshape(h, w))
            prm = R.make_params(w, h, **kw)
            want = oracle.render_bvh(*bunny_mesh, oracle.camera_turntable(w, h, t), _oracle_params(oracle, prm))
            common.assert_frames_equal(got, want, "batch t=%.2f %dx%d" % (t, w, h))
    # a frame with nothing on screen (scene behind the eye) and one without a depth attachment among ordinary ones
    w, h = 640, 360
    behind = R.camera_reference(w, h)
    behind.view[14] = 5.0
    cams = [R.camera_reference(w, h), behind, R.camera_turntable(w, h, 1.5)]
    prm = R.make_params(w, h, flags=R.LAMBERT)
    bufs = [torch.full((3, h * w), 0x5A5A5A5A, dtype=torch.int32, device="cuda") for _ in cams]
    torch.cuda.synchronize()
    scene.render_batch(scene.make_batch(cams, [prm] * 3, [(b[0].data_ptr(), 0 if i == 2 else b[1].data_ptr(), b[2].data_ptr())
                                                          for i, b in enumerate(bufs)]), 0)
    ctx.frame_wait(0)
    ocams = [oracle.camera_reference(w, h), oracle.Camera.from_buffer_copy(bytes(behind)), oracle.camera_turntable(w, h, 1.5)]
    for i, (b, ocam) in enumerate(zip(bufs, ocams)):
        host = b.cpu().numpy()
        want = oracle.render_bvh(*bunny_mesh, ocam, _oracle_params(oracle, prm))
        assert np.array_equal(host[0].view(np.uint8).reshape(h, w, 4), want[0]), "colour, frame %d" % i
        assert np.array_equal(host[2].view(np.uint32).reshape(h, w), want[2]), "primitive id, frame %d" % i
        if i == 2:
            assert (host[1] == 0x5A5A5A5A).all(), "a missing depth attachment must not be written"
        else:
            assert np.array_equal(host[1].view(np.uint32).reshape(h, w), want[1].view(np.uint32)), "depth, frame %d" % i
    assert (bufs[1][2].cpu().numpy().view(np.uint32) == 0xFFFFFFFF).all()
    # a bad frame anywhere in the batch refuses the whole batch before anything is drawn
    bufs = [torch.zeros(3, 64 * 64, dtype=torch.int32, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    bad = scene.make_batch([R.camera_reference(64, 64)] * 2, [R.make_params(64, 64), R.make_params(64, 64, tile=12)],
                           [(b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr()) for b in bufs])
    with pytest.raises(R.InvertError):
        scene.render_batch(bad)
    ctx.synchronize()
    assert int(bufs[0].abs().sum()) == 0
    scene.render_batch(scene.make_batch([], [], []))  # an empty batch is nothing
    scene.close()
    ctx.close()


def test_frames_join_orders_the_context_stream(bunny_path):
    """inv_frames_join: a copy enqueued on the context's stream after the join sees the finished frames of every
    slot, with no host wait in between."""
    import torch
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream()
    ctx = R.Context(0, stream=stream.cuda_stream)
    scene = R.Scene(ctx, R.Mesh.load_obj(bunny_path))
    w, h = 1920, 1080
    cam, prm = R.camera_reference(w, h), R.make_params(w, h)
    assert ctx.frame_stream(0) != 0 and ctx.frame_stream(0) == ctx.frame_stream(0) != ctx.frame_stream(1)
    frames = [torch.zeros(3, h * w, dtype=torch.int32, device="cuda") for _ in range(2)]
    copies = [torch.zeros_like(f) for f in frames]
    torch.cuda.synchronize()
    for rep in range(8):  # the slots keep several draws queued
        for slot, f in enumerate(frames):
            scene.render_in_flight(slot, cam, prm, f[0].data_ptr(), f[1].data_ptr(), f[2].data_ptr())
    ctx.frames_join()
    with torch.cuda.stream(stream):
        for c, f in zip(copies, frames):
            c.copy_(f, non_blocking=True)
    stream.synchronize()  # only the context's stream
    want = common.golden_frame("bunny_1080_rect")
    for c in copies:
        host = c.cpu().numpy()
        got = (host[0].view(np.uint8).reshape(h, w, 4), host[1].view(np.float32).reshape(h, w),
               host[2].view(np.uint32).reshape(h, w))
        common.assert_frames_equal(got, want, "after the join")
    scene.close()
    ctx.close()


def test_pinned_mesh_builds_the_same_scene(gpu_ctx, bunny_path):
    """inv_mesh_pin: the arrays move to page-locked memory (accessors keep working, pinning twice is harmless) and a
    scene built from them draws the same frame; so does a second build from the same pinned mesh."""
    mesh = R.Mesh.load_obj(bunny_path)
    v0, i0 = mesh.vertices, mesh.indices
    mesh.pin().pin()
    assert mesh.counts() == (2503, 14904)
    assert np.array_equal(mesh.vertices.view(np.uint32)[:, :3], v0.view(np.uint32)[:, :3]) and np.array_equal(mesh.indices, i0)
    for _ in range(2):
        scene = R.Scene(gpu_ctx, mesh)
        got = scene.render(R.camera_reference(512, 512), R.make_params(512, 512))
        common.assert_frames_equal(got, common.golden_frame("bunny_512"), "pinned mesh")
        scene.close()
    empty = R.Mesh.from_arrays(np.zeros((0, 4), np.float32), np.zeros(0, np.uint32)).pin()
    assert empty.counts() == (0, 0)
    mesh.close()
    empty.close()


def test_build_scratch_is_kept_grown_and_trimmed(oracle, gpu_ctx, bunny_mesh):
    """A context keeps the memory a build needs only while it runs: builds of growing and shrinking size, with the
    memory released in between, give the same scenes as ever (a smaller build runs inside the larger one's memory)."""
    xyzw, idx = bunny_mesh
    mesh = R.Mesh.from_arrays(xyzw, idx)
    small = R.Mesh.from_arrays(xyzw, idx[:3 * 700])
    tr = common.grid_translations(3)
    cam, prm = R.camera_reference(512, 512), R.make_params(512, 512)
    want_small = oracle.render_bvh(xyzw, idx[:3 * 700], oracle.camera_reference(512, 512), _oracle_params(oracle, prm))
    far = common.pulled_back_camera(R.camera_reference(320, 200), 0.27 * 3)
    fx, fi = common.flatten_instances(xyzw, idx, tr)
    prm_far = R.make_params(320, 200)
    want_grid = oracle.render_bvh(fx, fi, oracle.Camera.from_buffer_copy(bytes(far)), _oracle_params(oracle, prm_far))
    gpu_ctx.trim()  # nothing kept yet, or whatever an earlier test left
    for step in ("small", "bunny", "grid", "small", "trim", "bunny", "grid"):
        if step == "trim":
            gpu_ctx.trim()
            gpu_ctx.trim()  # twice is fine
            continue
        if step == "grid":
            scene = R.Scene(gpu_ctx, mesh, translations=tr)
            common.assert_frames_equal(scene.render(far, prm_far), want_grid, step)
        elif step == "small":
            scene = R.Scene(gpu_ctx, small)
            common.assert_frames_equal(scene.render(cam, prm), want_small, step)
        else:
            scene = R.Scene(gpu_ctx, mesh)
            common.assert_frames_equal(scene.render(cam, prm), common.golden_frame("bunny_512"), step)
        scene.close()
    mesh.close()
    small.close()


def test_headless_host_program(oracle, bunny_path, bunny_mesh, tmp_path):
    """tools/headless/invert_headless (C++ on the C ABI: load, build, frame loop with frames in flight, attachments
    written out): the files of the last frame against the golden (parity mode) and the oracle (turntable)."""
    import json
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "headless", "invert_headless")
    if not os.path.exists(exe):
        pytest.skip("tools/headless/invert_headless is not built (python -c 'import __graft_entry__ as g; g.build()')")

    def run(*args):
        p = subprocess.run([exe, "--obj", bunny_path, *args], capture_output=True, text=True, timeout=100)
        assert p.returncode == 0, p.stdout + p.stderr
        return json.loads(p.stdout.strip().splitlines()[-1])

    def read(prefix, w, h):
        raw = open(prefix + ".ppm", "rb").read()
        head = ("P6\n%d %d\n255\n" % (w, h)).encode()
        rgb = np.frombuffer(raw[len(head):], np.uint8).reshape(h, w, 3)
        raw = open(prefix + ".pfm", "rb").read()
        head = ("Pf\n%d %d\n-1.0\n" % (w, h)).encode()
        depth = np.frombuffer(raw[len(head):], "<f4").reshape(h, w)[::-1]
        primid = np.frombuffer(open(prefix + ".pid", "rb").read()[16:], "<u4").reshape(h, w)
        return rgb, depth, primid

    out = str(tmp_path / "parity")
    r = run("--width", "512", "--height", "512", "--frames", "5", "--in-flight", "2", "--per-launch", "2", "--out", out)
    assert r["triangles"] == 4968 and r["frames"] == 5 and r["hit_pixels_last_frame"] == 5651
    rgb, depth, primid = read(out, 512, 512)
    color, gdepth, gprimid = common.golden_frame("bunny_512")
    assert np.array_equal(rgb, color[..., [2, 1, 0]]) and np.array_equal(primid, gprimid)
    assert np.array_equal(depth.view(np.uint32), gdepth.view(np.uint32))
    from PIL import Image
    assert np.array_equal(np.asarray(Image.open(out + ".png")), color[..., [2, 1, 0, 3]])

    out = str(tmp_path / "turn")
    w, h, frames = 640, 360, 7
    r = run("--width", str(w), "--height", str(h), "--frames", str(frames), "--in-flight", "3", "--per-launch", "1",
            "--turntable", "--lambert", "--invert", "--pin", "--out", out)
    assert r["flags"] == (R.LAMBERT | R.INVERT) and r["us_per_frame"] > 0
    rgb, depth, primid = read(out, w, h)
    want = oracle.render_bvh(*bunny_mesh, oracle.camera_turntable(w, h, (frames - 1) / 60.0),
                             oracle.make_params(w, h, flags=R.LAMBERT | R.INVERT))
    assert np.array_equal(rgb, want[0][..., [2, 1, 0]]) and np.array_equal(primid, want[2])
    assert np.array_equal(depth.view(np.uint32), want[1].view(np.uint32))


def test_8k_frame_properties(bunny_scene, gpu_ctx):
    """7680x4320 (33 M pixels, far beyond what the oracle renders in seconds): the 8x8-tile variant, the 16x16-block
    variant and the per-lane traversal must agree bit for bit, the background must hold the clear values, and the
    covered area must be 16 times the 1080p one to within the silhouette's edge pixels."""
    import torch
    w, h = 7680, 4320
    cam = R.camera_reference(w, h)
    bufs = {}
    for name, flags in (("tiles", 0), ("blocks", R.BLOCKS), ("lanes", R.LANE_TRAVERSAL)):
        b = torch.empty(3, h * w, dtype=torch.int32, device="cuda")
        bunny_scene.render_device(cam, R.make_params(w, h, flags=flags | R.LAMBERT), b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr())
        bufs[name] = b
    gpu_ctx.synchronize()
    assert torch.equal(bufs["tiles"], bufs["blocks"]) and torch.equal(bufs["tiles"], bufs["lanes"])
    t = bufs["tiles"]
    miss = t[2] == -1
    hits = int((~miss).sum())
    assert abs(hits - 16 * 25139) < 0.01 * 16 * 25139
    assert bool((t[1][miss] == 0x3F800000).all()) and bool((t[0][miss] == -16777216).all())  # depth 1.0, colour (0,0,0,255)
    depth = t[1][~miss].view(torch.float32)
    assert float(depth.min()) > 0.90 and float(depth.max()) < 0.92  # the bunny's depth range (0.9028 .. 0.9150)


def test_ray_along_the_view_axis(oracle, bunny_scene, bunny_mesh):
    """Odd width and height: the centre pixel's ray has normalised coordinates (0, 0) and, with an unturned model, two
    direction components of exactly zero.  With the eye raised so that this ray hits the model, the box tests of the
    per-lane traversal (and of the blocks that fall back to it on so small a frame) have to cope with a direction
    whose reciprocal is infinite.  (Found by tools/fuzz_parity.py: the pixel came out as background.)"""
    for w, h, eye in ((81, 309, (0.043, 0.0619, -1.636)), (33, 33, (0.0, 0.11, -0.6)), (255, 129, (0.0, 0.1, -1.0))):
        cam = R.camera_reference(w, h)
        cam.view[12], cam.view[13], cam.view[14] = eye
        for flags in (0, R.BLOCKS, R.LANE_TRAVERSAL, R.BRUTE_FORCE):
            prm = R.make_params(w, h, flags=R.LAMBERT | R.INVERT | flags)
            got = bunny_scene.render(cam, prm)
            want = oracle.render_bvh(*bunny_mesh, oracle.Camera.from_buffer_copy(bytes(cam)), _oracle_params(oracle, prm))
            assert want[2][h // 2, w // 2] != 0xFFFFFFFF, "the centre ray is meant to hit"
            common.assert_frames_equal(got, want, "%dx%d flags %d" % (w, h, flags))
    # and occlusion rays: their directions come from a table and may have zero components as well
    cam = R.camera_reference(65, 65)
    cam.view[13], cam.view[14] = 0.11, -0.5
    prm = R.make_params(65, 65, ao_rays=16, flags=R.LAMBERT)
    want = oracle.render_bvh(*bunny_mesh, oracle.Camera.from_buffer_copy(bytes(cam)), _oracle_params(oracle, prm))
    common.assert_frames_equal(bunny_scene.render(cam, prm), want, "centre ray + occlusion")


def test_errors(bunny_scene):
    cam = R.camera_reference(64, 64)
    with pytest.raises(R.InvertError):
        bunny_scene.render(cam, R.make_params(64, 64, tile=12))
    with pytest.raises(R.InvertError):
        bunny_scene.render(cam, R.make_params(64, 64, flags=1 << 22))  # not an INV_* flag
    cam.proj[1] = 0.5  # not a GLM-style perspective
    with pytest.raises(R.InvertError) as e:
        bunny_scene.render(cam, R.make_params(64, 64))
    assert e.value.code == R.api.ERR_UNSUPPORTED


_check_lbvh = common.check_lbvh


def test_lbvh_structure_matches_the_sequential_restatement(oracle, bunny_scene, bunny_mesh):
    ntri, nnodes, nbytes = bunny_scene.counts()
    assert (ntri, nnodes) == (4968, 4967) and nbytes == 4968 * 48 + 4967 * 64
    _check_lbvh(oracle, bunny_scene, *bunny_mesh)


@pytest.mark.parametrize("ntri,seed", [(2, 0), (3, 1), (257, 2), (511, 9), (512, 10), (513, 11), (1023, 6), (1024, 7),
                                       (1025, 8), (2048, 3), (2049, 4), (20000, 5)])
def test_lbvh_structure_random(oracle, gpu_ctx, ntri, seed):
    """Sizes around the sort tile (2048 keys) and the refit's chunk of leaves (512), and duplicated triangles (equal
    Morton keys, index tie-break)."""
    xyz, faces = objfmt.random_soup(ntri, seed=seed, size=0.004)
    if ntri > 100:
        faces[ntri // 2:ntri // 2 + 40] = faces[:40]  # 40 exact duplicates
    xyzw = np.zeros((len(xyz), 4), np.float32)
    xyzw[:, :3] = xyz
    idx = faces.reshape(-1).astype(np.uint32)
    scene = R.Scene(gpu_ctx, R.Mesh.from_arrays(xyzw, idx))
    _check_lbvh(oracle, scene, xyzw, idx)
    prm = R.make_params(256, 256)
    got = scene.render(R.camera_reference(256, 256), prm)
    want = oracle.render_bvh(xyzw, idx, oracle.camera_reference(256, 256), _oracle_params(oracle, prm))
    common.assert_frames_equal(got, want, "%d triangles" % ntri)
    scene.close()


@pytest.mark.parametrize("copies", [47, 48, 49, 50, 97, 700])
@pytest.mark.parametrize("where", ["first", "middle", "last"])
def test_lbvh_runs_of_equal_keys(oracle, gpu_ctx, copies, where):
    """One triangle many times over (equal keys, ordered by index), at the low end, in the middle and at the high end
    of the sorted order: the build's short sorts order runs of up to 48 keys that share their upper bits in place
    and the build sorts again with more passes for longer ones; both sides of that limit, and runs that touch the ends
    of the array.  (A context remembers which sort a scene of so many triangles needed: every case has its own
    triangle count, so every case starts with the shortest.)"""
    ntri = 3000 + copies + {"first": 0, "middle": 1000, "last": 2000}[where]
    xyz, faces = objfmt.random_soup(ntri, seed=77, size=0.004)
    xyzw = np.zeros((len(xyz), 4), np.float32)
    xyzw[:, :3] = xyz
    order = oracle.lbvh_build(xyzw, faces.reshape(-1).astype(np.uint32))["order"]
    src = int(order[{"first": 0, "middle": ntri // 2, "last": ntri - 1}[where]])
    rng = np.random.default_rng(copies)
    dst = rng.choice(np.setdiff1d(np.arange(ntri), [src]), copies - 1, replace=False)
    faces[dst] = faces[src]
    idx = faces.reshape(-1).astype(np.uint32)
    scene = R.Scene(gpu_ctx, R.Mesh.from_arrays(xyzw, idx))
    _check_lbvh(oracle, scene, xyzw, idx)
    scene.close()


def test_lbvh_structure_large_instanced_scene(oracle, gpu_ctx, bunny_mesh):
    """12 x 12 copies (715 392 triangles: 350 sort tiles on all XCDs, 1 400 refit chunks): sorted
    keys, order, hierarchy and every box against the sequential restatement; then the same after a smaller build has
    used the context's scratch memory in between."""
    xyzw, idx = bunny_mesh
    tr = common.grid_translations(12)
    fx, fi = common.flatten_instances(xyzw, idx, tr)
    mesh = R.Mesh.from_arrays(xyzw, idx)
    for _ in range(2):
        scene = R.Scene(gpu_ctx, mesh, translations=tr)
        _check_lbvh(oracle, scene, fx, fi)
        scene.close()
        R.Scene(gpu_ctx, mesh).close()
    mesh.close()


@pytest.mark.parametrize("ntri,size", [(700, 0.02), (6000, 0.03)])
def test_tile_kernel_list_overflow_paths(oracle, gpu_ctx, ntri, size):
    """Dense overlapping soups at 64x64: more candidates per 8x8 tile than the record list (256) and, for the larger
    soup, than the range list holds, so the tile kernel works in batches or hands tiles to the per-lane traversal."""
    xyz, faces = objfmt.random_soup(ntri, seed=9, lo=(-0.02, 0.08, -0.02), hi=(0.02, 0.12, 0.02), size=size)
    xyzw = np.zeros((len(xyz), 4), np.float32)
    xyzw[:, :3] = xyz
    idx = faces.reshape(-1).astype(np.uint32)
    scene = R.Scene(gpu_ctx, R.Mesh.from_arrays(xyzw, idx))
    for spp in (1, 3):
        prm = R.make_params(64, 64, spp=spp, seed=3, flags=R.LAMBERT)
        got = scene.render(R.camera_reference(64, 64), prm)
        want = oracle.render_bvh(xyzw, idx, oracle.camera_reference(64, 64), _oracle_params(oracle, prm))
        common.assert_frames_equal(got, want, "%d triangles spp=%d" % (ntri, spp))
    scene.close()


def test_few_large_overlapping_triangles_overflow_the_lists(oracle, gpu_ctx):
    """Six hundred triangles as large as the scene, all over each other: few triangles per pixel of the rectangle (so
    the un-batched variants are chosen) but every 8x8 tile and 16x16 block meets more of them than its record list
    holds (128 / 256): tiles and blocks hand their pixels to the per-lane traversal."""
    xyz, faces = objfmt.random_soup(600, seed=21, lo=(-0.01, 0.10, -0.01), hi=(0.01, 0.12, 0.01), size=0.08)
    xyzw = np.zeros((len(xyz), 4), np.float32)
    xyzw[:, :3] = xyz
    idx = faces.reshape(-1).astype(np.uint32)
    scene = R.Scene(gpu_ctx, R.Mesh.from_arrays(xyzw, idx))
    w, h = 320, 240
    for flags, spp in ((0, 1), (R.BLOCKS, 1), (R.BLOCKS | R.LAMBERT, 2), (R.LAMBERT, 4)):
        prm = R.make_params(w, h, spp=spp, seed=5, flags=flags)
        got = scene.render(R.camera_reference(w, h), prm)
        want = oracle.render_bvh(xyzw, idx, oracle.camera_reference(w, h), _oracle_params(oracle, prm))
        common.assert_frames_equal(got, want, "large triangles flags=%d spp=%d" % (flags, spp))
    assert (got[2] != 0xFFFFFFFF).sum() > 1500
    scene.close()


def test_camera_inside_the_scene(oracle, gpu_ctx):
    """Eye inside the mesh's bounding box: the projected rectangle degenerates to the whole frame, boxes straddle
    the eye plane."""
    xyz, faces = objfmt.random_soup(500, seed=12, lo=(-0.3, -0.3, 0.7), hi=(0.3, 0.3, 1.3), size=0.08)
    xyzw = np.zeros((len(xyz), 4), np.float32)
    xyzw[:, :3] = xyz
    idx = faces.reshape(-1).astype(np.uint32)
    scene = R.Scene(gpu_ctx, R.Mesh.from_arrays(xyzw, idx))
    for mode in (0, R.LANE_TRAVERSAL, R.BLOCKS):
        prm = R.make_params(200, 150, flags=mode)
        got = scene.render(R.camera_reference(200, 150), prm)
        want = oracle.render_bruteforce(xyzw, idx, oracle.camera_reference(200, 150), _oracle_params(oracle, prm))
        common.assert_frames_equal(got, want, "camera inside")
    assert (got[2] != 0xFFFFFFFF).sum() > 1000
    scene.close()


@pytest.mark.parametrize("mode", MODES)
def test_ambient_occlusion_against_oracle(oracle, bunny_scene, bunny_mesh, mode):
    """Occlusion rays (any-hit, incoherent) on top of each render mode; colour is the only attachment they change."""
    w, h = 320, 240
    prm = R.make_params(w, h, ao_rays=16, seed=5, flags=R.LAMBERT | mode)
    got = bunny_scene.render(R.camera_reference(w, h), prm)
    want = oracle.render_bvh(*bunny_mesh, oracle.camera_reference(w, h), _oracle_params(oracle, prm))
    common.assert_frames_equal(got, want, "ao 16 rays mode %d" % mode)
    plain = bunny_scene.render(R.camera_reference(w, h), R.make_params(w, h, flags=R.LAMBERT | mode))
    hit = got[2] != 0xFFFFFFFF
    assert (got[0][hit].astype(int) <= plain[0][hit].astype(int)).all() and (got[0][hit] < plain[0][hit]).any()


@pytest.mark.parametrize("rays,flags,tmax", [(64, R.LAMBERT | R.INVERT, 0.0), (7, 0, 0.02), (100, R.RGBA, 0.0)])
def test_ambient_occlusion_ray_counts(oracle, bunny_scene, bunny_mesh, rays, flags, tmax):
    w, h = 400, 300
    prm = R.make_params(w, h, ao_rays=rays, seed=1, flags=flags, ao_tmax=tmax)
    got = bunny_scene.render(R.camera_reference(w, h), prm)
    oprm = _oracle_params(oracle, prm)
    oprm.ao_tmax = tmax
    want = oracle.render_bvh(*bunny_mesh, oracle.camera_reference(w, h), oprm)
    common.assert_frames_equal(got, want, "ao %d rays" % rays)


def test_ambient_occlusion_brute_force_oracle_small(oracle, bunny_scene, bunny_mesh):
    """The oracle's own brute-force any-hit (no hierarchy anywhere) at a size it finishes in seconds."""
    w, h = 160, 120
    prm = R.make_params(w, h, ao_rays=16, seed=5, flags=R.LAMBERT)
    got = bunny_scene.render(R.camera_reference(w, h), prm)
    want = oracle.render_bruteforce(*bunny_mesh, oracle.camera_reference(w, h), _oracle_params(oracle, prm))
    common.assert_frames_equal(got, want, "ao vs brute-force oracle")


def test_ambient_occlusion_config5_1080p(oracle, bunny_scene, bunny_mesh):
    """1080p, 64 occlusion rays per hit pixel (about 1.6 M secondary rays)."""
    prm = R.make_params(1920, 1080, ao_rays=64, flags=R.LAMBERT | R.INVERT)
    got = bunny_scene.render(R.camera_reference(1920, 1080), prm)
    want = oracle.render_bvh(*bunny_mesh, oracle.camera_reference(1920, 1080), _oracle_params(oracle, prm))
    common.assert_frames_equal(got, want, "config 5")


def test_ambient_occlusion_needs_one_sample_per_pixel(bunny_scene):
    with pytest.raises(R.InvertError) as e:
        bunny_scene.render(R.camera_reference(64, 64), R.make_params(64, 64, spp=2, ao_rays=8))
    assert e.value.code == R.api.ERR_UNSUPPORTED


@pytest.mark.parametrize("mode", MODES)
def test_instanced_scene_small(oracle, gpu_ctx, bunny_path, bunny_mesh, mode):
    """3x3 translated copies flattened into one LBVH; primitive id = instance * 4968 + face."""
    tr = common.grid_translations(3)
    scene = R.Scene(gpu_ctx, R.Mesh.load_obj(bunny_path), translations=tr)
    assert scene.counts()[0] == 9 * 4968
    fx, fi = common.flatten_instances(*bunny_mesh, tr)
    w, h = 480, 360
    cam = common.pulled_back_camera(R.camera_reference(w, h), 1.6)
    ocam = oracle.Camera.from_buffer_copy(bytes(cam))
    prm = R.make_params(w, h, flags=R.LAMBERT | mode)
    got = scene.render(cam, prm)
    want = oracle.render_bvh(fx, fi, ocam, _oracle_params(oracle, prm))
    common.assert_frames_equal(got, want, "3x3 instances")
    ids = got[2][got[2] != 0xFFFFFFFF]
    assert len(np.unique(ids // 4968)) == 9
    if mode == 0:
        _check_lbvh(oracle, scene, fx, fi)
    scene.close()


def test_instanced_scene_with_sub_pixel_triangles(oracle, gpu_ctx, bunny_path, bunny_mesh):
    """12x12 copies (715 392 triangles) seen from afar at 640x360: several triangles per pixel, so the draw takes the
    tile kernel that works through a tile's candidates in batches; also with jitter and with occlusion rays.  The
    build goes through the multi-block scan and many refit levels."""
    tr = common.grid_translations(12)
    scene = R.Scene(gpu_ctx, R.Mesh.load_obj(bunny_path), translations=tr)
    assert scene.counts()[0] == 144 * 4968
    fx, fi = common.flatten_instances(*bunny_mesh, tr)
    w, h = 640, 360
    cam = common.pulled_back_camera(R.camera_reference(w, h), 0.27 * 12)
    ocam = oracle.Camera.from_buffer_copy(bytes(cam))
    for kw in (dict(), dict(spp=3, seed=2), dict(ao_rays=8, seed=1)):
        prm = R.make_params(w, h, flags=R.LAMBERT | R.INVERT, **kw)
        got = scene.render(cam, prm)
        want = oracle.render_bvh(fx, fi, ocam, _oracle_params(oracle, prm))
        common.assert_frames_equal(got, want, "12x12 instances %r" % (kw,))
    hits = got[2] != 0xFFFFFFFF
    assert hits.sum() > 20000 and len(np.unique(got[2][hits] // 4968)) == 144
    scene.close()


def test_mesh_from_the_general_reader_renders(oracle, gpu_ctx, tmp_path):
    """A free-form OBJ with quads and i/t/n corners through the tolerant reader, rendered against the oracle."""
    xyz, faces = objfmt.icosphere(2, radius=0.06, center=(0.0, 0.11, 0.0))
    lines = ["v %r %r %r" % tuple(float(np.float32(c)) for c in p) for p in xyz]
    lines += ["vn 0 0 1", "vt 0 0"]
    lines += ["f %d/1/1 %d/1/1 %d/1/1" % tuple(int(i) + 1 for i in f) for f in faces]
    path = tmp_path / "sphere.obj"
    path.write_text("\r\n".join(lines))
    mesh = R.Mesh.load_obj_general(str(path))
    assert mesh.counts() == (len(xyz), faces.size)
    scene = R.Scene(gpu_ctx, mesh)
    w, h = 256, 256
    prm = R.make_params(w, h, flags=R.LAMBERT, ao_rays=8)
    got = scene.render(R.camera_reference(w, h), prm)
    want = oracle.render_bvh(mesh.vertices, mesh.indices, oracle.camera_reference(w, h), _oracle_params(oracle, prm))
    common.assert_frames_equal(got, want, "icosphere")
    assert (got[2] != 0xFFFFFFFF).sum() > 500
    scene.close()


def _deep_tree_scene():
    """Triangles whose Morton keys peel off one key bit per tree level: pairs of large triangles centred on points
    that close in on one interior point P along alternating axes (the cell of P with one coordinate bit flipped, from
    the top bit down), so the Karras tree over them is a chain some sixty levels deep, every level an internal node
    (the pair) beside the rest of the chain.  The triangles are large and all overlap around P: a ray through there
    meets both children at every level, and its stack of far children fills up."""
    lo = np.array([-0.08, 0.03, -0.08])  # the bounds the keys are quantised on (pinned by two tiny triangles below):
    ext = 0.16                            # every large triangle stays inside them
    g = np.array([0x0CCD93, 0x0D1EE5, 0x0C7B7B])  # P's grid cell, 21 bits per axis: about 0.4 of the way along each
    pts = []
    for j in range(20, 0, -1):
        for a in range(3):
            q = g.copy()
            q[a] ^= 1 << j
            pts.append(q)
    pts.append(g)
    xyz, faces = [], []

    def tri(c, s, flip):
        n = len(xyz)
        v = [(-s, -s, -s), (s, -s, s), (-s, s, s)] if not flip else [(s, s, s), (-s, s, -s), (s, -s, -s)]
        for d in v:
            xyz.append(c + np.array(d))
        faces.append((n, n + 1, n + 2) if not flip else (n, n + 2, n + 1))

    for k, q in enumerate(pts):
        c = lo + (q + 0.5) / 2.0 ** 21 * ext
        tri(c, 0.006 + 0.0003 * (k % 7), False)
        tri(c, 0.006 + 0.0003 * (k % 5), True)
    for corner in (lo, lo + ext):  # pin the bounds the keys are quantised on
        tri(corner + np.where(corner == lo, 1e-6, -1e-6), 1e-6, False)
    xyzw = np.ones((len(xyz), 4), np.float32)
    xyzw[:, :3] = np.array(xyz)
    return xyzw, np.array(faces, np.uint32).reshape(-1)


def _tree_height(child):
    depth = np.zeros(len(child), np.int64)
    order = [0]
    for i in order:  # parents before children
        for c in child[i]:
            if not (int(c) >> 31):
                depth[int(c)] = depth[i] + 1
                order.append(int(c))
    return int(depth.max()) + 1


def test_tree_deeper_than_the_traversal_stack(oracle, gpu_ctx):
    """An LBVH of more levels than the per-lane stacks hold (40): the tile and block kernels draw it like any other
    scene, and a lane whose stack fills up (their overflow path, INV_LANE_TRAVERSAL) tests the far subtree leaf by
    leaf; the occlusion kernel sizes its stacks by the tree's height."""
    xyzw, idx = _deep_tree_scene()
    scene = R.Scene(gpu_ctx, R.Mesh.from_arrays(xyzw, idx))
    common.check_lbvh(oracle, scene, xyzw, idx)
    assert _tree_height(scene.read_lbvh()["child"]) > 55
    w, h = 640, 480
    cam, ocam = R.camera_reference(w, h), oracle.camera_reference(w, h)
    assert scene.draw_path(cam, R.make_params(w, h)) == R.api.PATH_TILES
    for kw in (dict(flags=0), dict(flags=R.LANE_TRAVERSAL | R.LAMBERT), dict(flags=R.BLOCKS), dict(flags=R.LAMBERT, spp=4, seed=3),
               dict(flags=R.LAMBERT | R.INVERT, ao_rays=16), dict(flags=R.LANE_TRAVERSAL, ao_rays=5, spp=1)):
        prm = R.make_params(w, h, **kw)
        got = scene.render(cam, prm)
        want = oracle.render_bruteforce(xyzw, idx, ocam, _oracle_params(oracle, prm))
        common.assert_frames_equal(got, want, "deep tree %r" % (kw,))
    assert (got[2] != 0xFFFFFFFF).sum() > 200
    scene.close()

