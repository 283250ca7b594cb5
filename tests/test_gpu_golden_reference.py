"""The HIP path, called through the C ABI, against tests/golden/reference_kernels.npz: outputs of the REFERENCE's own kernels (its
.cu files compiled for the host, tests/make_golden_reference_kernels.py) for a three-keyframe scene, stage by stage -- depth
filter, keyframe preprocessing, surfel creation (plain and filtered), activation + geometry step, pose normal equations, deletion +
radius update, compaction, colour assignment, supporting surfels + merging, the PCG system, the intrinsics step.  The checks and their tolerances are the ones tests/test_cpu_golden_reference.py applies to the oracle
(tests/golden_reference.py); neither /root/reference nor the oracle is needed here.

Every test runs twice: with the exact arithmetic flavour of the sweeps (the default: the oracle's bits) and with the fast one
(bahip_context_set_arithmetic: v_rcp_f32 / v_sqrt_f32 / v_exp_f32, contraction, flushed denormals -- the arithmetic of the reference's
own -use_fast_math build).  The checks and tolerances are the same for both -- the fast flavour is held to the reference's kernels
as the exact one is -- with one exception: the 99.9th percentile of the position error after the geometry step (5e-7 m for the exact
flavour, measured 5.2e-7 m and held to 1e-6 m for the fast one; BASELINE's bar is 1e-5 m)."""
import numpy as np
import pytest

from badslam_amd import capi
from tests import golden_reference as gr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fix():
    return gr.load()


@pytest.fixture(scope="module", params=["exact", "fast"])
def arithmetic(request):
    return request.param


def _scene(fix, arithmetic, ctx=None):
    from badslam_amd import lowlevel as ll
    ctx = ctx or ll.Context()
    ctx.set_arithmetic(arithmetic)
    assert ctx.arithmetic == arithmetic
    cam, cam2 = ll.make_camera(fix["camera"], gr.WIDTH, gr.HEIGHT), ll.make_camera(fix["camera"], gr.WIDTH, gr.HEIGHT)
    return ll.Scene(ctx, gr.CAPACITY, float(fix["raw_to_float_depth"]), float(fix["baseline_fx"]), gr.CELL, cam, cam2)


def test_depth_filter(fix):
    from badslam_amd import lowlevel as ll
    ctx = ll.Context()
    s = float(fix["raw_to_float_depth"])
    for k in range(gr.KEYFRAMES):
        got = ll.bilateral_filtering_and_depth_cutoff(ctx, fix["raw"][k], *gr.BILATERAL, int(gr.MAX_DEPTH_M / s), s)
        gr.check_filtered(got, fix, k)


def _scene_with_reference_images(fix, check, arithmetic):
    """Keyframes built by the backend's own preprocessing from the reference's filtered depth (with `check`: held against the
    reference's keyframe images on the way), then given the reference's images so that every later stage starts from the file's state."""
    g = _scene(fix, arithmetic)
    for k in range(gr.KEYFRAMES):
        g.add_keyframe(fix["filtered"][k], fix["rgb"][k], fix["poses"][k])
        kf = g.keyframes[k]
        if check:
            rgba = kf["color"].download()
            assert np.array_equal(rgba[..., :3], fix["rgb"][k])
            gr.check_keyframe_images(kf["depth"].download(), kf["normals"].download(), kf["radius"].download(), rgba[..., 3], kf["min_depth"],
                                     kf["max_depth"], fix, k)
        kf["depth"].upload(np.ascontiguousarray(fix["depth"][k]))
        kf["normals"].upload(np.ascontiguousarray(fix["normals"][k]))
        kf["radius"].upload(np.ascontiguousarray(fix["radius"][k]))
        kf["color"].upload(np.ascontiguousarray(gr.rgba_of(fix, k)))
        kf["min_depth"], kf["max_depth"] = float(fix["min_max_depth"][k, 0]), float(fix["min_max_depth"][k, 1])
    g.bind_keyframes()
    return g


@pytest.fixture(scope="module")
def gpu(fix, arithmetic):
    return _scene_with_reference_images(fix, check=True, arithmetic=arithmetic)


def test_keyframe_preprocessing(gpu):
    assert len(gpu.keyframes) == gr.KEYFRAMES        # the checks ran while the fixture built the keyframes


def test_surfel_creation(fix, gpu):
    gpu.upload_surfels(np.zeros((8, 0), np.float32), np.zeros(0, np.uint8))
    counts = [gpu.create_surfels_for_keyframe(k, filter_new_surfels=False) for k in range(gr.KEYFRAMES)]
    gr.check_created(gpu.download_surfels()[:8], counts, fix)


def test_activation_and_geometry_step(fix, gpu, arithmetic):
    state = gr.perturbed_state(fix["created_rows"])
    n = state.shape[1]
    gpu.upload_surfels(state, np.zeros(n, np.uint8))
    for k, activation in enumerate(gr.ACTIVATIONS):
        gpu.keyframes[k]["activation"] = activation
    gpu.bind_keyframes()
    try:
        gpu.update_surfel_activation()
        active = gpu.active_buf.download()[0, :n].copy()
        gpu.optimize_geometry_iteration(True, True)
        rows = gpu.download_surfels()[:8].copy()
    finally:
        for k in range(gr.KEYFRAMES):
            gpu.keyframes[k]["activation"] = capi.KF_ACTIVE
        gpu.bind_keyframes()
    gr.check_activation_and_geometry(active, rows, state, fix, p999=5e-7 if arithmetic == "exact" else 1e-6)


@pytest.mark.parametrize("name,use_depth,use_desc", [("both", True, True), ("depth", True, False), ("desc", False, True)])
def test_pose_normal_equations(fix, gpu, name, use_depth, use_desc):
    state = gr.perturbed_state(fix["created_rows"])
    gpu.upload_surfels(state, np.ones(state.shape[1], np.uint8))
    gpu.bind_keyframes()
    H, b = gpu.accumulate_pose_coeffs(0, use_depth, use_desc, fix["pose_frame_T_global"])
    gr.check_pose_equations(H, b, name, fix)


def test_deletion_then_compaction(fix, gpu):
    state = gr.state_for_deletion(gr.perturbed_state(fix["created_rows"]))
    n = state.shape[1]
    gpu.upload_surfels(state, np.zeros(n, np.uint8))
    gpu.bind_keyframes()
    deleted = gpu.delete_surfels_and_update_radii(gr.MIN_OBSERVATIONS)
    gr.check_deletion(gpu.surfel_buf.download()[:8, :n], deleted, fix)
    rows, active = gr.reference_state_after_deletion(fix)       # compaction from the reference's own state: pure data movement
    gpu.upload_surfels(rows, active)
    gpu.surfel_count = n - int(fix["deleted_mask"].sum())
    gpu.compact_surfels(with_active=True)
    m = gpu.surfels_size
    assert m == n - int(fix["deleted_mask"].sum())
    assert np.array_equal(gr.digest(gpu.download_surfels()[:8], gpu.active_buf.download()[0, :m]), fix["compacted_digest"])


def test_filtered_surfel_creation(fix, gpu):
    gpu.upload_surfels(np.zeros((8, 0), np.float32), np.zeros(0, np.uint8))
    counts = [gpu.create_surfels_for_keyframe(k, filter_new_surfels=True, min_observation_count=gr.MIN_OBSERVATIONS) for k in range(gr.KEYFRAMES)]
    gr.check_created(gpu.download_surfels()[:8], counts, fix, prefix="filtered_")


def test_colour_assignment(fix, gpu):
    state = gr.perturbed_state(fix["created_rows"])
    gpu.upload_surfels(state, np.zeros(state.shape[1], np.uint8))
    gpu.bind_keyframes()
    gpu.assign_colors()
    gr.check_colours(gpu.download_surfels()[5].copy(), fix)


@pytest.mark.parametrize("merge", [False, True])
def test_supporting_surfels(fix, gpu, merge):
    n = fix["created_rows"].shape[1]
    gpu.upload_surfels(fix["created_rows"], np.zeros(n, np.uint8))
    gpu.bind_keyframes()
    planes, merged = gpu.determine_supporting_surfels(1, fix["frame_T_global"][1], merge=merge)
    mask = gpu.surfel_buf.download()[0, :n].view(np.uint32) == 0x7fffffff
    assert merged == int(mask.sum())
    gr.check_supporting(planes, mask, merge, fix)


def test_pcg_system(fix, arithmetic):
    g = _scene_with_reference_images(fix, check=False, arithmetic=arithmetic)           # its own scene: the call adopts poses and intrinsics
    state = gr.perturbed_state(fix["created_rows"])
    n = state.shape[1]
    g.upload_surfels(state, np.ones(n, np.uint8))
    for k in range(gr.KEYFRAMES):
        g.keyframes[k]["pose"] = np.asarray(fix["pcg_poses"][k], np.float32)
    g.bind_keyframes()
    cells = g.cf_w * g.cf_h
    U = 6 * (gr.KEYFRAMES - 1) + 3 * n + 5 + cells + 4
    g.pcg_iteration(optimize_poses=True, optimize_geometry=True, optimize_depth_intrinsics=True, optimize_color_intrinsics=True,
                    max_inner_iterations=0, gauge_keyframe=gr.GAUGE_KEYFRAME)
    gr.check_pcg_system(g.read_pcg_vector(0, U), g.read_pcg_vector(1, U), n, cells, fix)


def test_intrinsics_step(fix, arithmetic):
    g = _scene_with_reference_images(fix, check=False, arithmetic=arithmetic)           # its own scene: the step changes cameras and the cfactor image
    state = gr.perturbed_state(fix["created_rows"])
    g.upload_surfels(state, np.ones(state.shape[1], np.uint8))
    gr.miscalibrate(g)
    g.cfactor.upload(gr.miscalibrated_cfactor())
    g.set_intrinsics()
    g.bind_keyframes()
    cc, dc, a = g.optimize_intrinsics(True, True)
    gr.check_intrinsics_step([dc.fx, dc.fy, dc.cx, dc.cy], [cc.fx, cc.fy, cc.cx, cc.cy], a, g.cfactor.download(), fix)


def test_alternating_iterations_end_to_end(fix, arithmetic):
    """BASELINE's bar, HIP path against the reference's own kernels: keyframe poses and surfel positions after two alternating
    iterations (activation, geometry step, batched Gauss-Newton pose estimation)."""
    g = _scene_with_reference_images(fix, check=False, arithmetic=arithmetic)           # its own scene: the poses change
    state = gr.perturbed_state(fix["created_rows"])
    n = state.shape[1]
    g.upload_surfels(state, np.zeros(n, np.uint8))
    for k in range(gr.KEYFRAMES):
        g.keyframes[k]["pose"] = np.asarray(fix["pcg_poses"][k], np.float32)
    g.bind_keyframes()
    steps = 0
    poses = None
    for _ in range(gr.ALTERNATING_ITERATIONS):
        g.update_surfel_activation()
        g.optimize_geometry_iteration(True, True)
        poses, its, _, _ = g.estimate_keyframe_poses(True, True)
        steps += int(np.sum(its))
        for k in range(gr.KEYFRAMES):
            g.keyframes[k]["pose"] = np.asarray(poses[k], np.float32)
        g.bind_keyframes()
    gr.check_alternating_iterations(poses, g.download_surfels()[:3], steps, fix)
