"""The benchmark configurations at their stated parameters, through the C ABI, against the CPU oracle.

  config 3  bunny, 3840 x 2160, 16 jittered samples per pixel, seed 0: the whole frame, and the same frame as the
            shards of 2 / 4 / 8 ranks put together again on this GPU (dense and sparse layouts)
  config 4  32 x 32 = 1024 bunny instances flattened into one LBVH (5 087 232 triangles), 1080p: the LBVH against
            the sequential build, the frame against the oracle; and 118 x 118 = 13 924 instances (69 174 432
            triangles, the scale the configuration states) the same way
  config 5  bunny, 1080p, 64 occlusion rays per hit pixel: as the shards of 8 ranks, and from close by (318 k hit
            pixels, 20 M occlusion rays)

Bar: bit-exact primitive id, depth bits and colour bytes.
"""
import numpy as np
import pytest

import common
import objfmt
import renderer_amd as R

pytestmark = pytest.mark.gpu

W4K, H4K = 3840, 2160
FLAGS = R.LAMBERT | R.INVERT


@pytest.fixture(scope="module")
def bunny_scene(gpu_ctx, bunny_path):
    scene = R.Scene(gpu_ctx, R.Mesh.load_obj(bunny_path))
    yield scene
    scene.close()


@pytest.fixture(scope="module")
def config3_frame(oracle, bunny_scene, bunny_mesh):
    """The 4K / 16 spp frame from the oracle, checked once against the single-rank GPU frame."""
    prm = R.make_params(W4K, H4K, spp=16, seed=0, flags=FLAGS)
    want = oracle.render_bvh(*bunny_mesh, oracle.camera_reference(W4K, H4K), common.oracle_params(oracle, prm))
    return want


def test_config3_4k_16spp_against_oracle(bunny_scene, config3_frame):
    prm = R.make_params(W4K, H4K, spp=16, seed=0, flags=FLAGS)
    assert bunny_scene.draw_path(R.camera_reference(W4K, H4K), prm) == R.api.PATH_TILES
    got = bunny_scene.render(R.camera_reference(W4K, H4K), prm)
    common.assert_frames_equal(got, config3_frame, "config 3")
    hits = got[2] != 0xFFFFFFFF
    assert 100000 < int(hits.sum()) < 101500  # 100 580 from a double-precision rasteriser, give or take edge pixels
    # sixteen samples: edge pixels hold blends, so the blue channel takes many more values than at one sample
    assert len(np.unique(got[0][hits][:, 0])) > 150


@pytest.mark.parametrize("nranks,idle", [(2, 0), (4, 0), (8, 0), (3, 1), (4, 1), (8, 1)])
def test_config3_sparse_shards_reassemble(bunny_scene, gpu_ctx, config3_frame, nranks, idle):
    """`idle`: rank 0 takes no tiles (INV_ROOT_IDLE), the hit tiles are dealt to the other ranks."""
    got, rect, grid, n = common.sparse_composite_single_gpu(bunny_scene, gpu_ctx, R.camera_reference(W4K, H4K), W4K, H4K,
                                                            nranks, 32, spp=16, seed=0,
                                                            flags=FLAGS | (R.ROOT_IDLE if idle else 0))
    common.assert_frames_equal(got, config3_frame, "config 3, sparse shards of %d ranks" % nranks)
    assert n == -(-grid[2] * grid[3] // (nranks - idle)) * 32 * 32


@pytest.mark.parametrize("nranks,idle", [(2, 0), (4, 0), (8, 0), (4, 1)])
def test_config3_dense_shards_reassemble(bunny_scene, gpu_ctx, config3_frame, nranks, idle):
    got = common.dense_composite_single_gpu(bunny_scene, gpu_ctx, R.camera_reference(W4K, H4K), W4K, H4K, nranks, 64,
                                            spp=16, seed=0, flags=FLAGS | (R.ROOT_IDLE if idle else 0))
    common.assert_frames_equal(got, config3_frame, "config 3, dense shards of %d ranks" % nranks)


def test_config4_1024_instances(oracle, gpu_ctx, bunny_path, bunny_mesh):
    side = 32
    tr = common.grid_translations(side)
    scene = R.Scene(gpu_ctx, R.Mesh.load_obj(bunny_path), translations=tr)
    try:
        ntri, nnodes, _ = scene.counts()
        assert (ntri, nnodes) == (1024 * 4968, 1024 * 4968 - 1)
        fx, fi = common.flatten_instances(*bunny_mesh, tr)
        common.check_lbvh(oracle, scene, fx, fi)
        w, h = 1920, 1080
        cam = common.pulled_back_camera(R.camera_reference(w, h), 0.27 * side)
        ocam = oracle.Camera.from_buffer_copy(bytes(cam))
        prm = R.make_params(w, h, flags=FLAGS)
        # eleven triangles per pixel under the instances: drawn one lane per triangle
        assert scene.draw_path(cam, prm) == R.api.PATH_SCATTER
        got = scene.render(cam, prm)
        want = oracle.render_bvh(fx, fi, ocam, common.oracle_params(oracle, prm))
        common.assert_frames_equal(got, want, "config 4")
        # the same frame through the tile kernel (a binning pass, then 8x8 tiles in batches): two samples per pixel
        # are outside what the triangle pass takes, and sample 0 alone decides depth and primitive id
        prm2 = R.make_params(w, h, spp=2, seed=1, flags=FLAGS)
        assert scene.draw_path(cam, prm2) == R.api.PATH_TILES_BATCHED | R.api.PATH_BINNED
        got2 = scene.render(cam, prm2)
        assert np.array_equal(got2[2], want[2]) and np.array_equal(got2[1].view(np.uint32), want[1].view(np.uint32))
        want2 = oracle.render_bvh(fx, fi, ocam, common.oracle_params(oracle, prm2))
        common.assert_frames_equal(got2, want2, "config 4, two samples per pixel")
        hits = got[2] != 0xFFFFFFFF
        assert hits.sum() > 300000 and len(np.unique(got[2][hits] // 4968)) == 1024  # every instance is seen
        # one sample through the LBVH on request (INV_TILE_TRAVERSAL): the same frame
        prm_t = R.make_params(w, h, flags=FLAGS | R.TILE_TRAVERSAL)
        assert scene.draw_path(cam, prm_t) == R.api.PATH_TILES_BATCHED | R.api.PATH_BINNED
        common.assert_frames_equal(scene.render(cam, prm_t), want, "config 4 through the tile kernel")
        # the same frame as a batch of two on a frame slot (the form bench.py times for small scenes) and in parity mode
        prm0 = R.make_params(w, h)
        got0 = scene.render(cam, prm0)
        want0 = oracle.render_bvh(fx, fi, ocam, common.oracle_params(oracle, prm0))
        common.assert_frames_equal(got0, want0, "config 4, parity mode")
    finally:
        scene.close()
        gpu_ctx.trim()  # 0.6 GB of build scratch


@pytest.mark.timeout(900)
def test_config4_at_the_stated_scale_of_70_million_triangles(oracle, gpu_ctx, bunny_path, bunny_mesh):
    """BASELINE.json's fourth configuration says "~70M tris"; with the 4 968-triangle mesh of res/ that is 118 x 118 =
    13 924 instances, 69 174 432 triangles (4.4 GB of node records: offsets beyond 32 bits; more sort tiles than one
    workgroup can add up, so the scan's second level runs as a kernel of its own).  The LBVH against the sequential
    build, the frame against the oracle, and the tile path against both."""
    side = 118
    tr = common.grid_translations(side)
    scene = R.Scene(gpu_ctx, R.Mesh.load_obj(bunny_path), translations=tr)
    try:
        ntri, nnodes, _ = scene.counts()
        assert (ntri, nnodes) == (13924 * 4968, 13924 * 4968 - 1) and ntri == 69174432
        fx, fi = common.flatten_instances(*bunny_mesh, tr)
        common.check_lbvh(oracle, scene, fx, fi)
        w, h = 1920, 1080
        cam = common.grid_camera(R.camera_reference(w, h), side)  # 31.9 away, far plane at 100
        assert cam.proj[10] != R.camera_reference(w, h).proj[10]
        ocam = oracle.Camera.from_buffer_copy(bytes(cam))
        prm = R.make_params(w, h, flags=FLAGS)
        assert scene.draw_path(cam, prm) == R.api.PATH_SCATTER  # some seventy triangles per pixel
        got = scene.render(cam, prm)
        want = oracle.render_bvh(fx, fi, ocam, common.oracle_params(oracle, prm))
        del fx, fi
        common.assert_frames_equal(got, want, "config 4 at 69 M triangles")
        hits = got[2] != 0xFFFFFFFF
        assert hits.sum() > 300000 and len(np.unique(got[2][hits] // 4968)) == 13924  # every instance is seen
        # through the LBVH: a binning pass and 8x8 tiles in batches; sample 0 alone decides depth and primitive id
        prm2 = R.make_params(w, h, spp=2, seed=1, flags=FLAGS)
        assert scene.draw_path(cam, prm2) == R.api.PATH_TILES_BATCHED | R.api.PATH_BINNED
        got2 = scene.render(cam, prm2)
        assert np.array_equal(got2[2], want[2]) and np.array_equal(got2[1].view(np.uint32), want[1].view(np.uint32))
    finally:
        scene.close()
        gpu_ctx.trim()  # some 4 GB of build scratch


def test_config5_ao64_as_shards_of_8_ranks(oracle, bunny_scene, gpu_ctx, bunny_mesh):
    w, h = 1920, 1080
    prm = R.make_params(w, h, ao_rays=64, flags=FLAGS)
    want = oracle.render_bvh(*bunny_mesh, oracle.camera_reference(w, h), common.oracle_params(oracle, prm))
    got, _, _, _ = common.sparse_composite_single_gpu(bunny_scene, gpu_ctx, R.camera_reference(w, h), w, h, 8, 32,
                                                      ao_rays=64, flags=FLAGS)
    common.assert_frames_equal(got, want, "config 5, sparse shards of 8 ranks")
    got = common.dense_composite_single_gpu(bunny_scene, gpu_ctx, R.camera_reference(w, h), w, h, 8, 64, ao_rays=64,
                                            flags=FLAGS)
    common.assert_frames_equal(got, want, "config 5, dense shards of 8 ranks")


def test_config5_ao64_from_close_by(oracle, bunny_scene, bunny_mesh):
    """Eye 0.3 from the model: the bunny fills the frame, 64 occlusion rays for each of some 318 000 hit pixels."""
    w, h = 1920, 1080
    cam = R.camera_reference(w, h)
    cam.view[12], cam.view[13], cam.view[14] = 0.0, 0.11, -0.3
    prm = R.make_params(w, h, ao_rays=64, flags=FLAGS)
    got = bunny_scene.render(cam, prm)
    want = oracle.render_bvh(*bunny_mesh, oracle.Camera.from_buffer_copy(bytes(cam)), common.oracle_params(oracle, prm))
    common.assert_frames_equal(got, want, "config 5 from close by")
    assert int((got[2] != 0xFFFFFFFF).sum()) > 300000
