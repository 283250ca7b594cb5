"""The HIP path against the committed outputs of the REFERENCE's own kernels for the whole chain from raw input at VGA (VERDICT r3,
item 2; tests/e2e_vga.py describes the chain, tests/make_golden_e2e_vga.py generates tests/golden/e2e_vga.npz): the same 20 raw
640x480 depth + RGB frames go through bahip_bilateral_filtering_and_depth_cutoff, the Keyframe constructor and
vis::DirectBA::BundleAdjustment(do_surfel_updates = true, three iterations, end tasks) -- no image of the reference is uploaded --
and the result is held against the reference with exact bilinear weights and with the texture unit's 8-bit weights: pose RMSE
<= 1e-5 m (BASELINE.json), the same number of surviving surfels (<= 0.1 % apart), 99 % of the reference's sampled surfels matched
within 1e-5 m.  The same chain by the oracle gives the HIP path's bits."""
import numpy as np
import pytest

from tests import e2e_vga as e2e

pytestmark = pytest.mark.gpu


def test_hip_path_from_raw_vga_input_matches_the_reference_kernels():
    from badslam_amd import lowlevel as ll
    from badslam_amd.directba import DirectBA
    with np.load(e2e.PATH) as f:
        golden = {name: f[name] for name in f.files}
    scene, raw, rgb, start = e2e.scene_and_raw_input()
    assert np.array_equal(e2e.input_digest(raw, rgb, start), golden["input_digest"]), "the regenerated raw input is not the generator's"
    K, s = raw.shape[0], scene.raw_to_float_depth
    ctx = ll.Context()
    ba = DirectBA(e2e.CAPACITY, s, scene.baseline_fx, e2e.CELL, e2e.WIDTH, e2e.HEIGHT, scene.camera, scene.camera,
                  surfel_merge_dist_factor=e2e.MERGE_FACTOR, min_observation_count=e2e.MIN_OBSERVATIONS)
    for k in range(K):
        filtered = ll.bilateral_filtering_and_depth_cutoff(ctx, raw[k], *e2e.BILATERAL, int(e2e.MAX_DEPTH_M / s), s)
        ba.AddKeyframe(filtered, rgb[k], start[k])
    # the frustum test of the host finds every pair of these keyframes co-visible: the lists the reference side was given
    assert [sorted(ba.keyframe_covisibility(k)) for k in range(K)] == e2e.all_pairs_covisibility(K)
    ba.SetSpatialSortCellSize(0)          # keep the lifecycle's own order (index-wise comparison with the oracle below)
    done, _ = ba.BundleAdjustment(do_surfel_updates=True, optimize_poses=True, optimize_geometry=True, min_iterations=e2e.ITERATIONS,
                                  max_iterations=e2e.ITERATIONS, increase_ba_iteration_count=True)
    assert done == e2e.ITERATIONS
    poses = np.asarray([ba.keyframe_pose(k) for k in range(K)], np.float64)
    rows = ba.download_surfels(8)
    assert rows.shape[1] == ba.surfel_count()
    for prefix in ("", "quantized_"):
        e2e.check(e2e.compare(poses, rows, golden, prefix))
    # and the oracle's replay of the chain (tests/test_cpu_e2e_vga.py) is the HIP path, bit for bit
    orc = e2e.run_oracle(scene, raw, rgb, start)
    assert orc["final_surfels"] == rows.shape[1]
    assert np.array_equal(np.asarray(orc["poses"], np.float32), poses.astype(np.float32))
    assert np.array_equal(orc["rows"].view(np.uint32), rows.view(np.uint32))


def _run_hip_chain(row_major):
    from badslam_amd import lowlevel as ll
    from badslam_amd.directba import DirectBA
    scene, raw, rgb, start = e2e.scene_and_raw_input()
    K, s = raw.shape[0], scene.raw_to_float_depth
    ctx = ll.Context()
    ba = DirectBA(e2e.CAPACITY, s, scene.baseline_fx, e2e.CELL, e2e.WIDTH, e2e.HEIGHT, scene.camera, scene.camera,
                  surfel_merge_dist_factor=e2e.MERGE_FACTOR, min_observation_count=e2e.MIN_OBSERVATIONS)
    ba.SetRowMajorCreation(row_major)
    for k in range(K):
        filtered = ll.bilateral_filtering_and_depth_cutoff(ctx, raw[k], *e2e.BILATERAL, int(e2e.MAX_DEPTH_M / s), s)
        ba.AddKeyframe(filtered, rgb[k], start[k])
    ba.SetSpatialSortCellSize(0)
    done, _ = ba.BundleAdjustment(do_surfel_updates=True, optimize_poses=True, optimize_geometry=True, min_iterations=e2e.ITERATIONS,
                                  max_iterations=e2e.ITERATIONS, increase_ba_iteration_count=True)
    assert done == e2e.ITERATIONS
    return np.asarray([ba.keyframe_pose(k) for k in range(K)], np.float64), ba.surfel_count()


def _pose_rmse(a, b):
    return float(np.sqrt(np.mean(np.sum((a[:, 4:] - b[:, 4:]) ** 2, axis=1)))), float(np.max(np.linalg.norm(a[:, 4:] - b[:, 4:], axis=1)))


def test_hip_path_against_the_unmodified_reference_order():
    """VERDICT r4 missing 2 / next 7a.  The golden's main record gives the reference's kernels THIS backend's tile-major append order;
    `rowmajor_poses` / `rowmajor_final_surfels` are the reference run with its own row-major order, nothing permuted
    (B/kernel_create_surfels.cu:357-390).  (1) The default HIP chain is held against that unmodified run: the poses agree far inside
    BASELINE's 1e-5 m although ~1 % of the merge survivors differ.  (2) DirectBA::SetRowMajorCreation(true) makes the backend append
    in the reference's order -- its indices, hence its survivors: the surfel count then equals the unmodified run's to within the
    0.1 % the other record reaches, and the poses move closer still."""
    with np.load(e2e.PATH) as f:
        ref_poses, ref_count = f["rowmajor_poses"].astype(np.float64), int(f["rowmajor_final_surfels"])
        tile_major_count = int(f["final_surfels"])
    poses_default, count_default = _run_hip_chain(False)
    rmse, worst = _pose_rmse(poses_default, ref_poses)
    print(f"tile-major HIP chain vs the unmodified reference run: pose RMSE {rmse:.2e} m (max {worst:.2e} m), surfels {count_default} vs {ref_count}")
    assert rmse <= 1e-5 and worst <= 1e-5, (rmse, worst)
    assert abs(count_default - tile_major_count) <= 1e-3 * tile_major_count
    poses_rm, count_rm = _run_hip_chain(True)
    rmse_rm, worst_rm = _pose_rmse(poses_rm, ref_poses)
    print(f"row-major HIP chain vs the unmodified reference run: pose RMSE {rmse_rm:.2e} m (max {worst_rm:.2e} m), surfels {count_rm} vs {ref_count}")
    assert rmse_rm <= 3e-6 and worst_rm <= 1e-5, (rmse_rm, worst_rm)
    assert abs(count_rm - ref_count) <= 1e-3 * ref_count, (count_rm, ref_count)
