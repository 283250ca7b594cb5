"""Edge cases of the path on the GPU: ragged image sizes (plane tiles and wavefronts with padding), a deleted keyframe in
the middle of the list, empty surfel sets, an all-invalid depth image, a single keyframe, capacity exceeded."""
import ctypes as C

import numpy as np
import pytest

from badslam_amd import synthetic
from tests import common

pytestmark = pytest.mark.gpu


def test_ragged_image_size_matches_oracle():
    """330 x 243: neither a multiple of the 8x4 plane tile, nor of the 64x4 pixel blocks, nor of the 8-cell creation tile."""
    scene = common.small_scene(num_keyframes=3, width=330, height=243, seed=21)
    ba = common.build_oracle(scene, 300000)
    g = common.build_gpu(scene, 300000)
    for k in range(3):
        arrs, kf = ba.kf_arrays(k), g.keyframes[k]
        for name in ("depth", "normals", "color"):
            assert np.array_equal(kf[name].download(), arrs[name]), (k, name)
    ref, active = common.oracle_surfels(ba)
    got = g.download_surfels()
    assert got.shape[1] == ref.shape[1] > 5000
    assert np.array_equal(got[:8].view(np.uint32), ref[:8].view(np.uint32))
    # one geometry step from a displaced cloud, both launch shapes
    rng = np.random.Generator(np.random.PCG64(3))
    data = ref.copy()
    data[2] += rng.uniform(0, 0.004, data.shape[1]).astype(np.float32)
    from badslam_amd import capi
    for tile_waves in (1, 4):
        capi.check(g.ctx.lib.bahip_debug_set_launch_shapes(tile_waves, 0))
        ba.surfel_data[:, :data.shape[1]] = data
        g.upload_surfels(data, active * 0)
        g.bind_keyframes()
        g.update_surfel_activation()
        ba.update_surfel_activation()
        g.optimize_geometry_iteration(True, True)
        ba.optimize_geometry_iteration()
        assert np.array_equal(g.download_surfels()[:8].view(np.uint32), ba.surfel_data[:8, :data.shape[1]].view(np.uint32))
    capi.check(g.ctx.lib.bahip_debug_set_launch_shapes(0, 0))


def test_deleted_keyframe_in_the_middle():
    """A deleted keyframe is a NULL entry of the reference's keyframe list and is left out of the bound table; the
    four-class sums count positions among the remaining keyframes on both sides."""
    scene = common.small_scene(num_keyframes=6, seed=23)
    ba = common.build_oracle(scene, 500000)
    g = common.build_gpu(scene, 500000)
    data, active = common.oracle_surfels(ba)
    rng = np.random.Generator(np.random.PCG64(4))
    data[2] += rng.uniform(0, 0.004, data.shape[1]).astype(np.float32)
    ba.surfel_data[:, :data.shape[1]] = data
    ba.delete_keyframe(2)
    del g.keyframes[2]
    g.upload_surfels(data, active * 0)
    g.bind_keyframes()
    g.update_surfel_activation()
    ba.update_surfel_activation()
    assert np.array_equal(g.active_buf.download()[0, :data.shape[1]], ba.active[:data.shape[1]])
    g.optimize_geometry_iteration(True, True)
    ba.optimize_geometry_iteration()
    assert np.array_equal(g.download_surfels()[:8].view(np.uint32), ba.surfel_data[:8, :data.shape[1]].view(np.uint32))


def _directba(scene, cap, **kw):
    from badslam_amd.directba import DirectBA
    return DirectBA(cap, scene.raw_to_float_depth, scene.baseline_fx, scene.cell, scene.width, scene.height, scene.camera, scene.camera, **kw)


def test_bundle_adjustment_without_surfels_and_with_one_keyframe():
    scene = common.small_scene(num_keyframes=2, width=160, height=120, seed=25)
    # min_observation_count 1: with the default 2, the end-of-scheme tasks delete every surfel of a single keyframe
    ba = _directba(scene, 100000, min_observation_count=1, min_observation_count_while_bootstrapping_1=1,
                   min_observation_count_while_bootstrapping_2=1)
    ba.AddKeyframe(scene.depth[0], scene.rgb[0], scene.poses_gt[0])
    # one keyframe, no surfels: H = b = 0, the pose stays (B/direct_ba_alternating.cc:148-151), nothing crashes
    for use_pcg in (False, True):
        done, _ = ba.BundleAdjustment(optimize_poses=True, optimize_geometry=True, min_iterations=2, max_iterations=2, use_pcg=use_pcg)
        assert done == 2 and ba.surfel_count() == 0
        assert np.allclose(ba.keyframe_pose(0), np.asarray(scene.poses_gt[0], np.float64), atol=1e-6)
    # surfels from the only keyframe, then BA with surfel updates: every surfel has a single observation
    ba.CreateSurfelsForKeyframe(0, filter_new_surfels=False)
    n = ba.surfel_count()
    assert n > 1000
    before = ba.download_surfels(3)
    assert ba.unsorted_surfels() == n
    ba.BundleAdjustment(do_surfel_updates=False, optimize_poses=True, optimize_geometry=True, min_iterations=2, max_iterations=2)
    after = ba.download_surfels(3)
    assert after.shape == before.shape and np.isfinite(after).all()
    # the end tasks of that call put the buffer into Morton order (the surfels were appended since the last reorder): the same
    # surfels, permuted -- and, the input being consistent, nothing was there to correct
    assert ba.unsorted_surfels() == 0
    order_before, order_after = np.lexsort(before), np.lexsort(after)
    assert np.abs(after[:, order_after] - before[:, order_before]).max() < 1e-4
    assert not np.array_equal(order_before, order_after)
    # ... and with the reorder switched off the reference's surfel order stays observable
    ba.SetSpatialSortCellSize(0)
    ba.CreateSurfelsForKeyframe(0, filter_new_surfels=False)       # (nothing new: every cell is supported)
    again = ba.download_surfels(3)
    ba.BundleAdjustment(do_surfel_updates=False, optimize_poses=True, optimize_geometry=True, min_iterations=1, max_iterations=1)
    assert np.abs(ba.download_surfels(3) - again).max() < 1e-4


def test_all_invalid_depth_creates_no_surfels():
    scene = common.small_scene(num_keyframes=1, width=160, height=120, seed=26)
    ba = _directba(scene, 100000)
    ba.AddKeyframe(np.full_like(scene.depth[0], 65535), scene.rgb[0], scene.poses_gt[0])
    ba.CreateSurfelsForKeyframe(0, filter_new_surfels=False)
    assert ba.surfel_count() == 0 and ba.surfels_size() == 0
    assert (ba.keyframe_image(0, "depth") & 0x8000).all()


def test_capacity_exceeded_is_a_soft_failure():
    """B/kernel_create_surfels.cc:162-165: not enough room -> an error is logged, no surfels are created, no crash."""
    scene = common.small_scene(num_keyframes=2, width=160, height=120, seed=27)
    probe = _directba(scene, 100000)
    probe.AddKeyframe(scene.depth[0], scene.rgb[0], scene.poses_gt[0])
    probe.CreateSurfelsForKeyframe(0)
    need = probe.surfel_count()
    assert need > 1000
    ba = _directba(scene, need // 2)
    ba.AddKeyframe(scene.depth[0], scene.rgb[0], scene.poses_gt[0])
    ba.CreateSurfelsForKeyframe(0)
    assert ba.surfel_count() == 0
    done, _ = ba.BundleAdjustment(min_iterations=1, max_iterations=1)
    assert done == 1


def test_covisibility_lists_regrown_on_a_live_context():
    """re-growing the co-visibility CSR buffer freed the tile-bounds and window buffers too and left their
    pointers dangling.  Bind, run a pose phase (allocates the tile bounds) and set a window, then hand over lists with more
    than the 1024 entries of slack, and run everything again: results must equal those of a fresh context."""
    from badslam_amd import capi
    scene = common.small_scene(num_keyframes=6, seed=31)
    rng = np.random.Generator(np.random.PCG64(4))
    perturbed = [common.synthetic.perturb_pose(rng, T) for T in scene.poses_gt]

    def phase(g, lists):
        K = len(g.keyframes)
        g.bind_keyframes()           # (the pose phase leaves its result in the device table: start every phase from the same poses)
        offsets = np.zeros(K + 1, np.int32)
        offsets[1:] = np.cumsum([len(l) for l in lists])
        indices = np.asarray([j for l in lists for j in l], np.int32)
        window = np.ones(K, np.uint8)
        capi.check(g.ctx.lib.bahip_set_covisibility(g.ctx.handle, offsets.ctypes.data_as(C.POINTER(C.c_int)),
                                                indices.ctypes.data_as(C.POINTER(C.c_int)), K))
        capi.check(g.ctx.lib.bahip_set_activation_window(g.ctx.handle, window.ctypes.data_as(C.POINTER(C.c_uint8)), K))
        poses, its, conv, rounds = g.estimate_keyframe_poses(True, True)
        return np.asarray(poses, np.float32), list(its)

    def fresh():
        g = common.build_gpu(scene, 300000)
        for k, T in enumerate(perturbed):
            g.keyframes[k]["pose"] = np.asarray(T, np.float32)
        g.bind_keyframes()
        return g

    K = len(scene.depth)
    short = [[j for j in range(K) if j != k] for k in range(K)]
    long_lists = [[j for j in range(K) if j != k] * 60 for k in range(K)]     # 6 * 300 = 1800 entries > 30 + 1024 slack
    g = fresh()
    first = phase(g, short)
    second = phase(g, long_lists)                                              # re-grows the CSR buffer on the live context
    third = phase(g, short)
    ref = fresh()
    assert np.array_equal(first[0], phase(ref, short)[0])
    ref2 = fresh()
    want = phase(ref2, long_lists)
    assert np.array_equal(second[0], want[0]) and second[1] == want[1]
    assert np.array_equal(third[0], first[0])
    del g, ref, ref2                                                           # destroying the contexts must not double-free
