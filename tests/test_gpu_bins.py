"""The per-frame binning pass: the scene's fine cut, and draws that start from their bins instead of the coarse cut.

A draw gives the same pixels whether it is binned or not (the winner is a minimum over a superset of candidates), so
every parity test applies to both; this file checks the fine cut itself, then runs the parity suite again in a process
that bins every draw (INV_BINS=1: by default only frames that share a launch and scenes with a capped coarse cut are
binned), and draws that mix binned and un-binned frames.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

import common
import objfmt
import renderer_amd as R

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FINE = 16  # INV_FINE_CUT_LEAVES


def _check_fine_cut(scene, xyzw, idx):
    """Every sorted leaf in exactly one record of at most FINE leaves, each box the union of its triangles' boxes."""
    lo, hi, first, count = scene.read_fine_cut()
    ntri = idx.size // 3
    assert len(first) > 0 and count.min() >= 1 and count.max() <= FINE
    order = np.argsort(first, kind="stable")
    f, c = first[order].astype(np.int64), count[order].astype(np.int64)
    assert f[0] == 0 and np.array_equal(f[1:], (f + c)[:-1]) and f[-1] + c[-1] == ntri  # a partition of [0, ntri)
    tri = xyzw[idx.reshape(-1, 3).astype(np.int64), :3]  # [ntri, 3 corners, xyz], primitive order
    # the records' e1 = v1 - v0 is rounded once at assembly and the boxes are built from v0, v0 + e1, v0 + e2
    v0 = tri[:, 0]
    b1 = (v0 + (tri[:, 1] - v0).astype(np.float32)).astype(np.float32)
    b2 = (v0 + (tri[:, 2] - v0).astype(np.float32)).astype(np.float32)
    tlo, thi = np.minimum(v0, np.minimum(b1, b2)), np.maximum(v0, np.maximum(b1, b2))
    sorted_prim = scene.read_lbvh()["order"].astype(np.int64)
    tlo, thi = tlo[sorted_prim], thi[sorted_prim]
    want_lo = np.minimum.reduceat(tlo, f, axis=0)
    want_hi = np.maximum.reduceat(thi, f, axis=0)
    assert np.array_equal(lo[order], want_lo) and np.array_equal(hi[order], want_hi)
    return len(first)


def test_fine_cut_of_the_bunny(bunny_path, bunny_mesh, gpu_ctx):
    scene = R.Scene(gpu_ctx, R.Mesh.load_obj(bunny_path))
    n = _check_fine_cut(scene, *bunny_mesh)
    assert 4968 // FINE <= n <= 4968
    scene.close()


@pytest.mark.parametrize("ntri,seed", [(17, 1), (513, 2), (20000, 3)])
def test_fine_cut_of_random_soups(gpu_ctx, ntri, seed):
    xyz, faces = objfmt.random_soup(ntri, seed=seed, size=0.004)
    if ntri > 100:
        faces[ntri // 2:ntri // 2 + 40] = faces[:40]  # exact duplicates: equal keys, a run the index bits split
    xyzw = np.zeros((len(xyz), 4), np.float32)
    xyzw[:, :3] = xyz
    idx = faces.reshape(-1).astype(np.uint32)
    scene = R.Scene(gpu_ctx, R.Mesh.from_arrays(xyzw, idx))
    _check_fine_cut(scene, xyzw, idx)
    scene.close()


def test_tiny_scenes_have_no_fine_cut(gpu_ctx):
    xyz, faces = objfmt.random_soup(FINE, seed=4, size=0.01)
    xyzw = np.zeros((len(xyz), 4), np.float32)
    xyzw[:, :3] = xyz
    scene = R.Scene(gpu_ctx, R.Mesh.from_arrays(xyzw, faces.reshape(-1).astype(np.uint32)))
    assert len(scene.read_fine_cut()[2]) == 0
    scene.close()


def test_fine_cut_of_the_large_instanced_scene(gpu_ctx, bunny_mesh):
    xyzw, idx = bunny_mesh
    tr = common.grid_translations(12)
    fx, fi = common.flatten_instances(xyzw, idx, tr)
    scene = R.Scene(gpu_ctx, R.Mesh.from_arrays(xyzw, idx), translations=tr)
    n = _check_fine_cut(scene, fx, fi)
    assert n > 2048  # more than the coarse cut holds: single draws of this scene are binned
    scene.close()


def test_records_that_touch_more_bins_than_a_workgroup_counts_in_lds(oracle, gpu_ctx):
    """The binning pass counts a workgroup's (record, bin) pairs per bin in a 512-entry table in LDS and lets a pair that
    finds no place there update the bin's counter itself.  Here 64 triangles as large as the scene sit among 20 000 small
    ones (a scene of that size is binned by default): the records that hold them touch every one of the few thousand
    bins of the frame, far more than the table has entries."""
    xyz, faces = objfmt.random_soup(20000, seed=21, size=0.004)
    big, _ = objfmt.random_soup(64, seed=22, lo=(-0.02, 0.09, -0.04), hi=(0.0, 0.13, 0.04), size=0.09)
    xyz = np.concatenate([xyz, big])
    faces = np.arange(len(xyz), dtype=np.uint32).reshape(-1, 3)
    xyzw = np.zeros((len(xyz), 4), np.float32)
    xyzw[:, :3] = xyz
    idx = faces.reshape(-1).astype(np.uint32)
    scene = R.Scene(gpu_ctx, R.Mesh.from_arrays(xyzw, idx))
    w = h = 1536
    cam, ocam = R.camera_reference(w, h), oracle.camera_reference(w, h)
    for flags in (0, R.TILE_TRAVERSAL):
        prm = R.make_params(w, h, flags=flags)
        assert scene.draw_path(cam, prm) & R.api.PATH_BINNED
        got = scene.render(cam, prm)
        want = oracle.render_bvh(xyzw, idx, ocam, common.oracle_params(oracle, prm))
        common.assert_frames_equal(got, want, "large and small triangles, flags=%d" % flags)
    assert (got[2] >= 60000).sum() > 100000  # the large ones cover a good part of the frame
    scene.close()


_PARITY_NAMES = ("config1 or config2 or shading or jittered or samples_in or odd_extents or moved_camera or soup or tiny "
                 "or sharding or sparse_composite or turntable or in_flight or batched_draws or overflow or "
                 "camera_inside or ambient or instanced or ray_along or 8k")


def _parity_suite_with(names=_PARITY_NAMES, **env_vars):
    env = dict(os.environ, **env_vars)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"),
                        os.path.join(ROOT, "tests", "test_gpu_configs.py"), "-x", "-q", "-m", "gpu", "-k", names,
                        "--timeout", "300", "--timeout-method", "thread", "-p", "no:cacheprovider"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail


def test_parity_suite_with_every_draw_binned():
    """The parity tests again with INV_BINS=1 (its own process: the library reads the variable once)."""
    _parity_suite_with(INV_BINS="1")


def test_parity_suite_with_one_lane_per_triangle():
    """The parity tests again with INV_SCATTER=1: every frame of one sample per pixel seen from outside the scene is
    drawn one lane per triangle into a visibility buffer (by default only frames whose triangles are far smaller than
    a pixel are), including shards of several ranks and frames with occlusion rays."""
    _parity_suite_with(INV_SCATTER="1")


def test_lbvh_suite_with_the_scan_kernel_between_the_sort_kernels():
    """The LBVH structure tests again with INV_SORT_SCAN_KERNEL=1: the second level of the radix sort's scan runs as a
    kernel of its own, as it does for scenes of more than eight million triangles (smaller ones leave it to the scatter
    kernel), here on the scenes of 0.7 and 5 million."""
    _parity_suite_with(names="lbvh_structure_large or config4_1024 or lbvh_runs", INV_SORT_SCAN_KERNEL="1")


def test_lbvh_suite_with_the_searching_build():
    """The LBVH structure tests again with INV_BUILD_SEARCH=1: every node finds its range by T. Karras's searches, the
    chunks only refit and the nodes across chunk boundaries are climbed through memory (the build before the chunks
    and the levels above them merged bottom-up; still the way of trees too unbalanced for that): the same trees."""
    _parity_suite_with(names="lbvh or config4_1024 or tree_deeper or tiny_scenes", INV_BUILD_SEARCH="1")


def test_lbvh_suite_when_units_hand_up_more_tops_than_they_may():
    """INV_BUILD_TOP_CAP=6: a chunk may hand up six unfinished subtrees (ordinary chunks have ten), so the bottom-up
    build gives up on every scene of more than a chunk and the build starts again the searching way, as it would for
    a tree too unbalanced for the arrays: the same trees, and scenes of a single chunk never get there."""
    _parity_suite_with(names="lbvh or config4_1024 or tree_deeper or tiny_scenes", INV_BUILD_TOP_CAP="6")


def test_parity_suite_from_the_coarse_cut():
    """The parity tests again with INV_FLAT=0: blocks of small scenes walk the coarse cut's two levels, as blocks of
    larger scenes do, instead of testing the whole fine cut themselves (the default up to 1024 records)."""
    _parity_suite_with(INV_FLAT="0")

