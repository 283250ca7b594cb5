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


def test_merge_keyframes_deletes_what_the_reference_rule_names():
    """DirectBA::MergeKeyframes (B/direct_ba.cc:251-338; VERDICT r4 next 7c: implemented since round 3, never called by a test).
    The rule, restated here from the reference: for every keyframe with a successor, the z-axis angle and the distance to the successor;
    pairs beyond 45 degrees or 0.3 m break the chain; keyframe i > 0 becomes a candidate with the sum of its two half distances
    (distance + angle * 0.5 / (pi / 2)); the approx_merge_count smallest are deleted in that order unless a neighbour has gone."""
    from badslam_amd import se3
    from badslam_amd.directba import DirectBA
    scene = common.small_scene(num_keyframes=9, width=160, height=120, seed=23)
    # poses along a line with hand-picked gaps: (1, 2, 3) and (5, 6, 7) are tight groups, 3 -> 4 is too far, 7 -> 8 turns too much
    gaps = [0.20, 0.03, 0.02, 0.45, 0.10, 0.015, 0.05, 0.10]
    poses, x = [], 0.0
    for k in range(9):
        rot = [0.0, 0.9 if k == 8 else 0.02 * k, 0.0]
        poses.append(se3.exp(np.array([x, 0.0, 0.0] + rot, np.float64)))
        if k < 8:
            x += gaps[k]
    ba = DirectBA(200000, scene.raw_to_float_depth, scene.baseline_fx, scene.cell, scene.width, scene.height, scene.camera, scene.camera)
    for k in range(9):
        ba.AddKeyframe(scene.depth[k], scene.rgb[k], poses[k])

    def expected(alive, count):
        R = [se3.quat_to_rot(np.asarray(ba.keyframe_pose(k), np.float64)[:4]) if alive[k] else None for k in range(9)]
        t = [np.asarray(ba.keyframe_pose(k), np.float64)[4:] if alive[k] else None for k in range(9)]
        cands, prev_half, prev_id = [], np.float32(0), 0
        for k in range(8):
            if not alive[k]:
                continue
            nxt = next((n for n in range(k + 1, 9) if alive[n]), None)
            if nxt is None:
                break
            angle = np.float32(np.arccos(np.float32(np.dot(R[k][:, 2], R[nxt][:, 2]))))
            if angle > np.float32(0.5 * np.pi / 2):
                continue
            dist = np.float32(np.linalg.norm(t[k] - t[nxt]))
            if dist > np.float32(0.3):
                continue
            half = np.float32(dist + np.float32(0.5 / (np.pi / 2)) * angle)
            if k > 0:
                cands.append((float(prev_half + half), prev_id, k, nxt))
            prev_half, prev_id = half, k
        cands.sort(key=lambda c: c[0])
        gone = []
        for _, a, b, c in cands[:count]:
            if alive[a] and alive[b] and alive[c] and a not in gone and b not in gone and c not in gone:
                gone.append(b)
        return gone

    alive = [True] * 9
    want = expected(alive, 2)
    assert len(want) >= 1 and 0 not in want
    ba.MergeKeyframes(2)
    got = [k for k in range(9) if not ba.keyframe_exists(k)]
    assert sorted(got) == sorted(want), (got, want)
    assert ba.keyframe_exists(0) and ba.keyframe_exists(8)
    # a second call works on the thinned sequence (deleted slots are skipped when looking for the successor)
    for k in got:
        alive[k] = False
    want2 = expected(alive, 10)
    ba.MergeKeyframes(10)
    got2 = [k for k in range(9) if alive[k] and not ba.keyframe_exists(k)]
    assert sorted(got2) == sorted(want2), (got2, want2)
    # and the backend still runs on what is left (bound table rebuilt from the surviving keyframes)
    left = [k for k in range(9) if ba.keyframe_exists(k)]
    for k in left:
        ba.CreateSurfelsForKeyframe(k, filter_new_surfels=False)
    done, _ = ba.BundleAdjustment(do_surfel_updates=False, optimize_poses=True, optimize_geometry=True, min_iterations=1, max_iterations=1,
                                  increase_ba_iteration_count=False)
    assert done == 1 and ba.surfel_count() > 0 and ba.ExportToPointCloud_count() == ba.surfel_count()
