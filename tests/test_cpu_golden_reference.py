"""The oracle against tests/golden/reference_kernels.npz: outputs of the REFERENCE's own kernels (its .cu files compiled for the
host, tests/make_golden_reference_kernels.py) for a three-keyframe scene, stage by stage -- depth filter, keyframe
preprocessing, surfel creation (plain and filtered), activation + geometry step, pose normal equations, deletion + radius update,
compaction, colour assignment, supporting surfels + merging, the PCG system, the intrinsics step.  Unlike
tests/test_cpu_oracle_vs_reference.py this needs neither /root/reference nor the library built from it: the file is committed.
tests/test_gpu_golden_reference.py replays the same file through the HIP path."""
import ctypes as C

import numpy as np
import pytest

from oracle import binding as ob
from tests import golden_reference as gr


@pytest.fixture(scope="module")
def fix():
    return gr.load()


def test_depth_filter(fix):
    s = float(fix["raw_to_float_depth"])
    for k in range(gr.KEYFRAMES):
        gr.check_filtered(ob.bilateral_filter_and_depth_cutoff(fix["raw"][k], *gr.BILATERAL, int(gr.MAX_DEPTH_M / s), s), fix, k)


def test_keyframe_preprocessing(fix):
    W, H, s = gr.WIDTH, gr.HEIGHT, float(fix["raw_to_float_depth"])
    L = ob.lib()
    cam = ob.make_camera(fix["camera"], W, H)
    cf = np.zeros(((H - 1) // gr.CELL + 1, (W - 1) // gr.CELL + 1), np.float32)
    dp = ob.DepthParams(0.0, s, float(fix["baseline_fx"]), gr.CELL, ob._ptr(cf, C.c_float), cf.shape[1], cf.shape[0])
    ptr = lambda x: x.ctypes.data_as(C.c_void_p)
    for k in range(gr.KEYFRAMES):
        after_normals, normals, radius, depth = (np.zeros((H, W), np.uint16) for _ in range(4))
        filtered, rgb, rgba = np.ascontiguousarray(fix["filtered"][k]), np.ascontiguousarray(fix["rgb"][k]), np.zeros((H, W, 4), np.uint8)
        L.orc_compute_normals(C.byref(cam), C.byref(dp), ptr(filtered), ptr(after_normals), ptr(normals))
        L.orc_compute_point_radii(C.byref(cam), C.c_float(s), ptr(after_normals), ptr(radius), ptr(depth))
        lo, hi = C.c_float(), C.c_float()
        L.orc_compute_min_max_depth(ptr(after_normals), W, H, C.c_float(s), C.byref(lo), C.byref(hi))
        L.orc_compute_brightness(ptr(rgb), W, H, ptr(rgba))
        assert np.array_equal(rgba[..., :3], rgb)
        gr.check_keyframe_images(depth, normals, radius, rgba[..., 3], lo.value, hi.value, fix, k)


@pytest.fixture(scope="module")
def oracle(fix):
    """The oracle holding the reference's keyframe images; the matrices it derives from the poses are the file's."""
    ba = gr.oracle_with_reference_images(fix)
    for k in range(gr.KEYFRAMES):
        assert np.array_equal(np.array(list(ba.keyframes[k].frame_T_global), np.float32), fix["frame_T_global"][k])
        assert np.array_equal(np.array(list(ba.keyframes[k].global_R_frame), np.float32), fix["global_R_frame"][k])
    return ba


def _load(ba, rows):
    n = rows.shape[1]
    ba.surfel_data[:] = 0
    ba.surfel_data[:8, :n] = rows
    ba.surfels.surfels_size = ba.surfels.surfel_count = n


def test_surfel_creation(fix, oracle):
    _load(oracle, np.zeros((8, 0), np.float32))
    counts = []
    for k in range(gr.KEYFRAMES):
        before = oracle.surfels_size
        oracle.create_surfels_for_keyframe(k, filter_new_surfels=False)
        counts.append(oracle.surfels_size - before)
    gr.check_created(oracle.surfel_data[:8, :oracle.surfels_size], counts, fix)


def test_activation_and_geometry_step(fix, oracle):
    state = gr.perturbed_state(fix["created_rows"])
    n = state.shape[1]
    _load(oracle, state)
    oracle.active[:] = 0
    for k, activation in enumerate(gr.ACTIVATIONS):
        oracle.keyframes[k].activation = activation
    oracle.use_depth = oracle.use_desc = 1
    oracle.update_surfel_activation()
    active = oracle.active[:n].copy()
    oracle.optimize_geometry_iteration()
    for k in range(gr.KEYFRAMES):
        oracle.keyframes[k].activation = ob.KF_ACTIVE
    gr.check_activation_and_geometry(active, oracle.surfel_data[:8, :n], state, fix)


@pytest.mark.parametrize("name,use_depth,use_desc", [("both", 1, 1), ("depth", 1, 0), ("desc", 0, 1)])
def test_pose_normal_equations(fix, oracle, name, use_depth, use_desc):
    _load(oracle, gr.perturbed_state(fix["created_rows"]))
    oracle.use_depth, oracle.use_desc = use_depth, use_desc
    H, b, count, _ = oracle.accumulate_pose_coeffs(0, frame_T_global=fix["pose_frame_T_global"], accumulate_double=False)
    oracle.use_depth = oracle.use_desc = 1
    assert count > 1000
    gr.check_pose_equations(H, b, name, fix)


def test_deletion_then_compaction(fix, oracle):
    state = gr.state_for_deletion(gr.perturbed_state(fix["created_rows"]))
    n = state.shape[1]
    _load(oracle, state)
    deleted = oracle.delete_surfels_and_update_radii(gr.MIN_OBSERVATIONS)
    gr.check_deletion(oracle.surfel_data[:8, :n], deleted, fix)
    rows, active = gr.reference_state_after_deletion(fix)       # compaction from the reference's own state: pure data movement
    _load(oracle, rows)
    oracle.active[:n] = active
    oracle.surfels.surfel_count = n - int(fix["deleted_mask"].sum())
    oracle.compact_surfels()
    m = oracle.surfels_size
    assert m == n - int(fix["deleted_mask"].sum())
    assert np.array_equal(gr.digest(oracle.surfel_data[:8, :m], oracle.active[:m]), fix["compacted_digest"])


def test_filtered_surfel_creation(fix, oracle):
    _load(oracle, np.zeros((8, 0), np.float32))
    counts = []
    for k in range(gr.KEYFRAMES):
        before = oracle.surfels_size
        oracle.create_surfels_for_keyframe(k, filter_new_surfels=True)
        counts.append(oracle.surfels_size - before)
    gr.check_created(oracle.surfel_data[:8, :oracle.surfels_size], counts, fix, prefix="filtered_")


def test_colour_assignment(fix, oracle):
    state = gr.perturbed_state(fix["created_rows"])
    _load(oracle, state)
    oracle.assign_colors()
    gr.check_colours(oracle.surfel_data[5, :state.shape[1]].copy(), fix)


@pytest.mark.parametrize("merge", [False, True])
def test_supporting_surfels(fix, oracle, merge):
    n = fix["created_rows"].shape[1]
    _load(oracle, fix["created_rows"])
    planes = oracle.determine_supporting_surfels(1, merge)
    gr.check_supporting(planes, oracle.surfel_data[0, :n].view(np.uint32) == 0x7fffffff, merge, fix)


def test_pcg_system(fix, oracle):
    state = gr.perturbed_state(fix["created_rows"])
    _load(oracle, state)
    for k in range(gr.KEYFRAMES):
        oracle.set_pose(k, fix["pcg_poses"][k])
        assert np.array_equal(np.array(list(oracle.keyframes[k].frame_T_global), np.float32), fix["pcg_frame_T_global"][k])
    oracle.use_depth = oracle.use_desc = 1
    try:
        r, M = oracle.pcg_assemble(True, True, True, True, gauge_keyframe=gr.GAUGE_KEYFRAME)
    finally:
        for k in range(gr.KEYFRAMES):
            oracle.set_pose(k, fix["poses"][k])
    gr.check_pcg_system(r, M, state.shape[1], oracle.cf_w * oracle.cf_h, fix)


def test_intrinsics_step(fix):
    ba = gr.oracle_with_reference_images(fix)                     # its own scene: the step changes cameras and the cfactor image
    _load(ba, gr.perturbed_state(fix["created_rows"]))
    gr.miscalibrate(ba)
    cc, dc, a = ba.optimize_intrinsics(True, True)
    gr.check_intrinsics_step([dc.fx, dc.fy, dc.cx, dc.cy], [cc.fx, cc.fy, cc.cx, cc.cy], a, ba.cfactor, fix)


def test_alternating_iterations_end_to_end(fix):
    ba = gr.oracle_with_reference_images(fix)
    state = gr.perturbed_state(fix["created_rows"])
    n = state.shape[1]
    _load(ba, state)
    poses = [np.asarray(T, np.float64) for T in fix["pcg_poses"]]
    for k in range(gr.KEYFRAMES):
        ba.set_pose(k, poses[k])
    ba.use_depth = ba.use_desc = 1
    steps = 0
    for _ in range(gr.ALTERNATING_ITERATIONS):
        ba.update_surfel_activation()
        ba.optimize_geometry_iteration()
        for k in range(gr.KEYFRAMES):
            estimate, its, _ = ba.estimate_frame_pose(k, poses[k])
            poses[k] = estimate.to_array()
            steps += its
        for k in range(gr.KEYFRAMES):
            ba.set_pose(k, poses[k])
    gr.check_alternating_iterations(poses, ba.surfel_data[:3, :n], steps, fix)
