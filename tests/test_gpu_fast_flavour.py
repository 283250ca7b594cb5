"""The FAST arithmetic flavour of the sweeps (bahip_context_set_arithmetic, DirectBA::SetFastArithmetic; ba_launch.h "Two arithmetic
flavours") held to the reference: v_rcp_f32 / v_sqrt_f32 / v_exp_f32, contraction and flushed denormals are what the reference's own
build computes with (applications/badslam/CMakeLists.txt:74: -use_fast_math), so its bar is the reference's kernels within
BASELINE.json's tolerance -- not the oracle's bits, which stay the exact flavour's bar.

  * tests/test_gpu_golden_reference.py runs every stage-level check of tests/golden_reference.py with both flavours (same tolerances);
  * here: the whole chain from raw VGA input against the reference's kernels (tests/e2e_vga.py, unchanged checks), the reference's
    twelve closed-loop tests, determinism and launch-shape invariance of the fast flavour (its sums keep their defined order), and
    that the switch really selects other code;
  * tests/test_gpu_scale_parity.py::test_c3_fast_flavour_against_the_exact_build: BASELINE configs[2], pose RMSE and association
    flips of the fast flavour against the bit-exact build."""
import os
import subprocess

import numpy as np
import pytest

from badslam_amd import capi
from tests import common
from tests import e2e_vga as e2e

pytestmark = pytest.mark.gpu

BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "badslam_amd", "lib", "test_directba")


def _run_e2e_chain(fast):
    from badslam_amd import lowlevel as ll
    from badslam_amd.directba import DirectBA
    scene, raw, rgb, start = e2e.scene_and_raw_input()
    K, s = raw.shape[0], scene.raw_to_float_depth
    ctx = ll.Context()
    ba = DirectBA(e2e.CAPACITY, s, scene.baseline_fx, e2e.CELL, e2e.WIDTH, e2e.HEIGHT, scene.camera, scene.camera,
                  surfel_merge_dist_factor=e2e.MERGE_FACTOR, min_observation_count=e2e.MIN_OBSERVATIONS)
    ba.SetFastArithmetic(fast)
    for k in range(K):
        filtered = ll.bilateral_filtering_and_depth_cutoff(ctx, raw[k], *e2e.BILATERAL, int(e2e.MAX_DEPTH_M / s), s)
        ba.AddKeyframe(filtered, rgb[k], start[k])
    ba.SetSpatialSortCellSize(0)
    done, _ = ba.BundleAdjustment(do_surfel_updates=True, optimize_poses=True, optimize_geometry=True, min_iterations=e2e.ITERATIONS,
                                  max_iterations=e2e.ITERATIONS, increase_ba_iteration_count=True)
    assert done == e2e.ITERATIONS
    poses = np.asarray([ba.keyframe_pose(k) for k in range(K)], np.float64)
    rows = ba.download_surfels(8)
    assert rows.shape[1] == ba.surfel_count()
    return poses, rows


def test_fast_flavour_from_raw_vga_input_matches_the_reference_kernels():
    """The e2e golden (outputs of the reference's own kernels for the chain from raw 640x480 input through BundleAdjustment with the
    surfel lifecycle) with the fast flavour: tests/e2e_vga.py's checks, unchanged -- pose RMSE <= 1e-5 m (BASELINE.json; <= 3e-6
    reached), surfel counts within 0.1 %, 98.5 % of the sampled surfels within 1e-5 m -- against the reference with exact and with
    8-bit bilinear weights; and the distance to the exact flavour's own result, for the record."""
    with np.load(e2e.PATH) as f:
        golden = {name: f[name] for name in f.files}
    poses, rows = _run_e2e_chain(True)
    for prefix in ("", "quantized_"):
        e2e.check(e2e.compare(poses, rows, golden, prefix))
    exact_poses, exact_rows = _run_e2e_chain(False)
    rmse = float(np.sqrt(np.mean(np.sum((poses[:, 4:] - exact_poses[:, 4:]) ** 2, axis=1))))
    print(f"fast vs exact flavour, same chain: pose RMSE {rmse:.2e} m, surfels {rows.shape[1]} vs {exact_rows.shape[1]}")
    assert rmse <= 1e-5
    assert abs(rows.shape[1] - exact_rows.shape[1]) <= 1e-3 * exact_rows.shape[1]
    assert not np.array_equal(poses.astype(np.float32), exact_poses.astype(np.float32)), "the fast flavour produced the exact flavour's bits: was it selected?"


@pytest.mark.parametrize("name", ["PoseOptimizationWithGeometricResidual", "PoseOptimizationColorOnlyCues",
                                  "AlternatingGeometryOptimizationWithGeometricResidual",
                                  "PCGGeometryOptimizationWithGeometricResidual",
                                  "AlternatingGeometryOptimizationWithPhotometricResidual",
                                  "PCGGeometryOptimizationWithPhotometricResidual",
                                  "AlternatingIntrinsicsOptimizationWithPhotometricResidual",
                                  "PCGIntrinsicsOptimizationWithPhotometricResidual",
                                  "AlternatingDepthDeformationOptimizationWithGeometricResidual",
                                  "PCGDepthDeformationOptimizationWithGeometricResidual",
                                  "AlternatingIntrinsicsOptimizationWithGeometricResidual",
                                  "PCGIntrinsicsOptimizationWithGeometricResidual"])
def test_reference_closed_loop_with_the_fast_flavour(name):
    """The reference's twelve closed-loop BA tests (B/test/test_*.cc restated in badslam_amd/host/test_directba.cc, with the
    reference's own bounds) with every context of the process defaulting to the fast flavour (BAHIP_ARITHMETIC=fast)."""
    assert os.path.exists(BIN), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    proc = subprocess.run([BIN, name], capture_output=True, text=True, timeout=600, env=dict(os.environ, BAHIP_ARITHMETIC="fast"))
    print(proc.stdout)
    print(proc.stderr)
    assert proc.returncode == 0, proc.stdout[-2000:]


def _iteration(g, perturbed):
    for k, T in enumerate(perturbed):
        g.keyframes[k]["pose"] = np.asarray(T, np.float32)
    g.bind_keyframes()
    g.update_activation_and_optimize_geometry(True, True)
    poses, its, conv, rounds = g.estimate_keyframe_poses(True, True)
    return np.asarray(poses, np.float64), g.download_surfels()[:8].copy(), g.active_buf.download()[0, :g.surfels_size].copy()


def test_fast_flavour_is_deterministic_and_launch_shape_invariant():
    """What the flavour does NOT give up: the order of every sum.  Two runs, and runs with other launch shapes (one / four wavefronts
    per tile in the geometry step, the pose sweep's work items split over 1 / 8 wavefronts, LDS / global-atomic form), give the same
    bits -- the property that makes a sharded fast run the unsharded fast run."""
    from badslam_amd import lowlevel as ll, synthetic
    scene = common.small_scene(num_keyframes=6, seed=5)
    rng = np.random.Generator(np.random.PCG64(2))
    perturbed = [synthetic.perturb_pose(rng, T) for T in scene.poses_gt]
    ctx = ll.Context()
    ctx.set_arithmetic("fast")
    g = common.build_gpu(scene, 400000, ctx=ctx)
    data = g.download_surfels().copy()
    n = data.shape[1]
    lib = ctx.lib
    results = []
    try:
        for tile_waves, pose_parts, pose_form in ((0, 0, 0), (0, 0, 0), (1, 1, 1), (4, 8, 1), (4, 0, 2), (1, 0, 2)):
            capi.check(lib.bahip_debug_set_launch_shapes(tile_waves, pose_parts))
            capi.check(lib.bahip_debug_set_pose_form(pose_form))
            g.upload_surfels(data, np.zeros(n, np.uint8))
            results.append(_iteration(g, perturbed))
    finally:
        capi.check(lib.bahip_debug_set_launch_shapes(0, 0))
        capi.check(lib.bahip_debug_set_pose_form(0))
    for poses, rows, active in results[1:]:
        assert np.array_equal(poses, results[0][0])
        assert np.array_equal(rows.view(np.uint32), results[0][1].view(np.uint32))
        assert np.array_equal(active, results[0][2])
    # ... and it is other code than the exact flavour's: close, not identical
    ctx.set_arithmetic("exact")
    g.upload_surfels(data, np.zeros(n, np.uint8))
    exact_poses, exact_rows, exact_active = _iteration(g, perturbed)
    flips = int(np.count_nonzero(exact_active != results[0][2]))
    dpos = np.abs(exact_rows[:3] - results[0][1][:3]).max(axis=0)
    dt = np.linalg.norm(exact_poses[:, 4:] - results[0][0][:, 4:], axis=1)
    print(f"fast vs exact, one iteration on {n} surfels: {flips} activation flips, positions p99.9 {np.percentile(dpos, 99.9):.1e} m, poses max {dt.max():.1e} m")
    assert flips <= 1e-3 * n
    assert np.percentile(dpos, 99.9) < 1e-5 and dt.max() < 1e-5
    assert not np.array_equal(exact_rows.view(np.uint32), results[0][1].view(np.uint32))


def test_arithmetic_switch_is_validated():
    from badslam_amd import lowlevel as ll
    ctx = ll.Context()
    assert ctx.arithmetic == ("fast" if os.environ.get("BAHIP_ARITHMETIC") == "fast" else "exact")
    with pytest.raises(capi.BackendError):
        ctx.set_arithmetic(7)
    ctx.set_arithmetic("fast")
    assert ctx.arithmetic == "fast"
    ctx.set_arithmetic("exact")
    assert ctx.arithmetic == "exact"
