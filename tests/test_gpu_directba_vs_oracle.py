"""End-to-end parity: DirectBA::BundleAdjustment of the C++ host driver (HIP backend) against the
oracle's restatement of the same driver, on identical input.  Gates from BASELINE.md: keyframe pose
RMSE <= 1e-5 m, surfel positions <= 1e-5 m on the surfels whose association did not flip, flips
<= 0.1 %."""
import numpy as np
import pytest

from badslam_amd import synthetic
from tests import common

pytestmark = pytest.mark.gpu


def _build(scene, cap, use_depth=True, use_desc=True, min_obs=2):
    from badslam_amd.directba import DirectBA
    ba = DirectBA(cap, scene.raw_to_float_depth, scene.baseline_fx, scene.cell, scene.width, scene.height, scene.camera,
                  scene.camera, use_depth_residuals=use_depth, use_descriptor_residuals=use_desc,
                  min_observation_count_while_bootstrapping_1=min_obs, min_observation_count_while_bootstrapping_2=min_obs,
                  min_observation_count=min_obs)
    for k in range(len(scene.depth)):
        ba.AddKeyframe(scene.depth[k], scene.rgb[k], scene.poses_gt[k])
    return ba


@pytest.fixture(scope="module")
def scene():
    return common.small_scene(num_keyframes=6, seed=17)


def _translation_rmse(a, b):
    return float(np.sqrt(np.mean([np.sum((np.asarray(x[4:]) - np.asarray(y[4:])) ** 2) for x, y in zip(a, b)])))


def test_keyframe_images_and_surfels_identical(scene):
    orc = common.build_oracle(scene, 600000)
    ba = _build(scene, 600000)
    for k in range(len(scene.depth)):
        ba.CreateSurfelsForKeyframe(k, filter_new_surfels=False)
        for name in ("depth", "normals", "color"):
            assert np.array_equal(ba.keyframe_image(k, name), orc.kf_arrays(k)[name]), (k, name)
    ref, _ = common.oracle_surfels(orc)
    got = ba.download_surfels(8)
    assert got.shape == ref[:8].shape
    assert np.array_equal(got.view(np.uint32), ref[:8].view(np.uint32))


@pytest.mark.parametrize("use_pcg", [False, True])
def test_bundle_adjustment_fixed_surfels(scene, use_pcg):
    rng = np.random.Generator(np.random.PCG64(5))
    orc = common.build_oracle(scene, 600000)
    ba = _build(scene, 600000)
    data, _ = common.oracle_surfels(orc)
    data[2] += rng.uniform(0, 0.004, data.shape[1]).astype(np.float32)
    orc.surfel_data[:, :data.shape[1]] = data
    ba.upload_surfels(data[:8])
    perturbed = [synthetic.perturb_pose(rng, T) for T in scene.poses_gt]
    for k, T in enumerate(perturbed):
        orc.set_pose(k, T)
        ba.set_keyframe_pose(k, T)
    iters = 3 if use_pcg else 5
    # vis::DirectBA re-establishes the Morton order of the buffer in its end tasks (default grid 0.02 m); the oracle does the
    # same when told to (its default is the reference's behaviour: never)
    orc.spatial_sort_cell, orc.unsorted_surfels = 0.02, ba.unsorted_surfels()
    cost_before, _ = orc.evaluate_cost()
    ba.set_pcg_gauge_keyframe(0)
    # increase_ba_iteration_count = True on both sides: the end-of-scheme tasks (surfel deletion, radius
    # update, compaction) run once after the loop
    done, conv = ba.BundleAdjustment(min_iterations=iters, max_iterations=iters, use_pcg=use_pcg, increase_ba_iteration_count=True,
                                     optimize_poses=True, optimize_geometry=True)
    stats = orc.bundle_adjustment(min_iterations=iters, max_iterations=iters, use_pcg=use_pcg, increase_ba_iteration_count=True,
                                  optimize_poses=True, optimize_geometry=True, pcg_gauge_keyframe=0)
    assert done == stats.iterations_done == iters
    K = len(perturbed)
    got_poses = [ba.keyframe_pose(k) for k in range(K)]
    ref_poses = [orc.pose(k) for k in range(K)]
    cost_after, _ = orc.evaluate_cost()
    # BA did its job (gauge-free measure; the photometric terms have a quantisation floor)
    assert cost_after < 0.8 * cost_before, (cost_before, cost_after)
    if not use_pcg:
        # alternating scheme: BASELINE.md gates
        assert _translation_rmse(got_poses, ref_poses) <= 1e-5, _translation_rmse(got_poses, ref_poses)
        # end-of-scheme tasks ran on both sides (increase_ba_iteration_count): same survivors
        assert ba.surfel_count() == orc.surfels_size
        got = ba.download_surfels(8)
        ref = orc.surfel_data[:8, :orc.surfels_size]
        dpos = np.abs(got[:3] - ref[:3]).max(axis=0)
        flips = np.count_nonzero(dpos > 1e-5)
        assert flips <= 1e-3 * ref.shape[1], (flips, np.quantile(dpos, [0.5, 0.99, 1.0]))
        # those are the north-star gates; every stage of the alternating scheme is a defined computation on both sides
        # (geometry: four-class sums; poses: tile tree + fixed point, binary64 solve, defined sin / cos), so five
        # iterations plus the end-of-scheme tasks end in the same bits
        assert np.array_equal(np.asarray(got_poses, np.float32), np.asarray(ref_poses, np.float32))
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    else:
        # PCG.  Every dense sum and dot product of the scheme is an exact sum on both sides (exact_sum.h / oracle_exact.c), the
        # per-surfel entries are ordered chains and the pose update uses the defined sin / cos: three outer iterations with
        # their inner conjugate-gradient loops plus the end-of-scheme tasks end in the same bits -- which contains the
        # north-star gates (pose RMSE <= 1e-5 m, surfel positions <= 1e-5 m) with room to spare.  The reference itself is not
        # reproducible run to run here (binary32 atomics, B/kernel_pcg.cu:98-154).
        assert _translation_rmse(got_poses, ref_poses) <= 1e-5
        assert ba.surfel_count() == orc.surfels_size
        got = ba.download_surfels(8)
        ref = orc.surfel_data[:8, :orc.surfels_size]
        assert np.count_nonzero(np.abs(got[:3] - ref[:3]).max(axis=0) > 1e-5) == 0
        assert np.array_equal(np.asarray(got_poses, np.float32), np.asarray(ref_poses, np.float32))
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_bundle_adjustment_with_surfel_updates(scene):
    """do_surfel_updates = true: creation (filtered), merging, deletion and compaction inside BA."""
    orc = common.build_oracle(scene, 600000, create_from=[])
    ba = _build(scene, 600000)
    rng = np.random.Generator(np.random.PCG64(9))
    perturbed = [synthetic.perturb_pose(rng, T, 0.002, 0.0005) for T in scene.poses_gt]
    for k, T in enumerate(perturbed):
        orc.set_pose(k, T)
        ba.set_keyframe_pose(k, T)
    orc.covis = [ba.keyframe_covisibility(k) for k in range(len(perturbed))]     # the host's frustum-based lists
    orc.spatial_sort_cell, orc.unsorted_surfels = 0.02, ba.unsorted_surfels()   # the end tasks reorder on both sides
    for call in range(2):
        done, _ = ba.BundleAdjustment(do_surfel_updates=True, min_iterations=2, max_iterations=2, increase_ba_iteration_count=True)
        orc.bundle_adjustment(do_surfel_updates=True, min_iterations=2, max_iterations=2, increase_ba_iteration_count=True)
        # every lifecycle stage is bit-exact against the oracle (test_gpu_lifecycle_stages.py) and the geometry step is
        # too, so the same surfels survive: exact counts
        assert ba.surfel_count() == orc.surfels_size, (call, ba.surfel_count(), orc.surfels_size)
    assert orc.surfels_size > 10000
    K = len(perturbed)
    got_poses = [ba.keyframe_pose(k) for k in range(K)]
    ref_poses = [orc.pose(k) for k in range(K)]
    assert _translation_rmse(got_poses, ref_poses) <= 1e-5
    assert np.array_equal(np.asarray(got_poses, np.float32), np.asarray(ref_poses, np.float32))
    got, ref = ba.download_surfels(8), orc.surfel_data[:8, :orc.surfels_size]
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
