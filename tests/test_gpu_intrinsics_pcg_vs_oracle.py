"""GPU parity: intrinsics step (Schur complement) and the PCG scheme vs the CPU oracle.

Per-pair terms are bit-identical on both sides (test_gpu_kernels_vs_oracle.py), and every sum over many of them is a DEFINED
sum on both sides: binary32 chains per surfel, fixed trees per 64-surfel tile, and then binary64 accumulation (intrinsics step)
or exact accumulation (PCG: exact_sum.h / oracle_exact.c) over the tiles.  So the kernels -- which merge with atomics in
arbitrary order -- and the oracle agree to the last bit, the number of inner conjugate-gradient steps included."""
import numpy as np
import pytest

from tests import common

pytestmark = pytest.mark.gpu


def _cam_tuple(c):
    return np.array([c.fx, c.fy, c.cx, c.cy], dtype=np.float64)


@pytest.fixture(scope="module")
def scene():
    return common.small_scene(num_keyframes=5, seed=21)


@pytest.fixture(params=[0, 1], ids=["records added by LDS f64 atomics", "records sorted by cell in LDS"])
def reduce_form(request):
    """Both forms of the second kernel of the intrinsics step (kernels_intrinsics.hip: intrinsics_bin_reduce_kernel /
    intrinsics_bin_reduce_sorted_kernel) against the oracle."""
    from badslam_amd import capi
    lib = capi.load()
    capi.check(lib.bahip_debug_set_intrinsics_reduce_form(request.param))
    yield request.param
    capi.check(lib.bahip_debug_set_intrinsics_reduce_form(-1))


def _perturbed_pair(scene, depth_cam_offset=(0.5, -0.6, 1.23, -2.17), color_cam_offset=(0, 0, 0, 0)):
    """Oracle + GPU scenes with identical surfels and identically perturbed cameras."""
    ba = common.build_oracle(scene, 400000)
    g = common.build_gpu(scene, 400000, create_from=[])
    data, active = common.oracle_surfels(ba)
    g.upload_surfels(data, np.ones_like(active))
    ba.active[:data.shape[1]] = 1
    for name, off in (("depth_cam", depth_cam_offset), ("color_cam", color_cam_offset)):
        for obj in (ba, g):
            cam = getattr(obj, name)
            cam.fx += off[0]; cam.fy += off[1]; cam.cx += off[2]; cam.cy += off[3]
    g.set_intrinsics()
    g.bind_keyframes()
    return ba, g


def _bits(values):
    return np.asarray(values, np.float32).view(np.uint32)


def test_depth_intrinsics_step(scene, reduce_form):
    """One depth-intrinsics + deformation step (Schur complement over the sparse cfactor cells).  The accumulation is DEFINED
    (binary32 terms, per-surfel chains, xor butterfly per tile, binary64 across tiles and per cell, xor butterfly + ordered
    partials in the Schur complement: kernels_intrinsics.hip / oracle_intrinsics.c), so kernels and oracle agree to the last
    bit although the kernels merge with atomics."""
    ba, g = _perturbed_pair(scene)
    ba.use_depth, ba.use_desc = 1, 1
    before = _cam_tuple(ba.depth_cam)
    cc_r, dc_r, a_r = ba.optimize_intrinsics(True, False)
    cc_g, dc_g, a_g = g.optimize_intrinsics(True, False)
    true_cam = np.asarray(scene.camera, np.float64)
    # one step removes most of the 0.5 .. 2.2 px perturbation (this also pins the oracle's Schur solve)
    assert np.abs(_cam_tuple(dc_r) - true_cam).max() < 0.1
    assert np.array_equal(_bits(_cam_tuple(dc_g)), _bits(_cam_tuple(dc_r))), (_cam_tuple(dc_g), _cam_tuple(dc_r))
    assert np.array_equal(_bits([a_g]), _bits([a_r])), (a_g, a_r)
    cf_r, cf_g = ba.cfactor, g.cfactor.download()
    assert np.count_nonzero(cf_r) > 0.5 * cf_r.size
    assert np.array_equal(_bits(cf_g), _bits(cf_r)), np.abs(cf_g - cf_r).max()

    # further steps with a != 0 (the first step leaves a at 0: with cfactor == 0 the residuals do not depend on it): the
    # deformation exp(-a / depth) now enters every residual -- a defined exponential on both sides (ba_device.h: exp_det), so
    # still every bit
    ba.dp.a = 0.0125
    g.dp.a = 0.0125
    g.set_intrinsics()
    for _ in range(2):
        _, dc_r2, a_r2 = ba.optimize_intrinsics(True, False)
        _, dc_g2, a_g2 = g.optimize_intrinsics(True, False)
        assert a_r2 != 0 and a_r2 != np.float32(0.0125)
        assert np.array_equal(_bits(_cam_tuple(dc_g2)), _bits(_cam_tuple(dc_r2))), (_cam_tuple(dc_g2), _cam_tuple(dc_r2))
        assert np.array_equal(_bits([a_g2]), _bits([a_r2])), (a_g2, a_r2)
        assert np.array_equal(_bits(g.cfactor.download()), _bits(ba.cfactor))


@pytest.mark.parametrize("capacity", [-1, 0, 16, 256])
def test_depth_intrinsics_step_whatever_the_record_buffers_hold(scene, capacity, reduce_form):
    """The sweep appends its per-cell records to per-block buffers and a second kernel adds them up in LDS; records that find
    their buffer full go out as atomics (kernels_intrinsics.hip).  Automatic size (-1), no buffers (0), buffers far too small
    (16 records per buffer: nearly everything overflows) and too small for the busier buffers only (256): the same bits."""
    import ctypes as C
    ba, g = _perturbed_pair(scene)
    ba.use_depth, ba.use_desc = 1, 1
    lib, h = g.ctx.lib, g.ctx.handle
    assert lib.bahip_debug_set_intrinsics_bin_capacity(h, capacity) == 0
    _, dc_r, a_r = ba.optimize_intrinsics(True, True)
    _, dc_g, a_g = g.optimize_intrinsics(True, True)
    cap, most, total = C.c_uint32(), C.c_uint32(), C.c_uint64()
    assert lib.bahip_debug_intrinsics_bin_stats(h, C.byref(cap), C.byref(most), C.byref(total)) == 0
    if capacity == 0:
        assert cap.value == 0
    else:
        assert total.value > 100000 and most.value > 256          # the scene's pairs went through the reservation
        assert (most.value > cap.value) == (capacity > 0), (cap.value, most.value)
    assert np.array_equal(_bits(_cam_tuple(dc_g)), _bits(_cam_tuple(dc_r))), (_cam_tuple(dc_g), _cam_tuple(dc_r))
    assert np.array_equal(_bits([a_g]), _bits([a_r]))
    assert np.array_equal(_bits(g.cfactor.download()), _bits(ba.cfactor))
    if capacity == -1:
        # the next call sizes its buffers from this call's demand
        g.optimize_intrinsics(True, True)
        assert lib.bahip_debug_intrinsics_bin_stats(h, C.byref(cap), C.byref(most), C.byref(total)) == 0
        assert most.value <= cap.value


@pytest.mark.parametrize("slices", [2, 3, 8, 16])
def test_depth_intrinsics_step_in_slices(scene, slices):
    """Round 5: on a large cloud the sweep runs in slices of its schedule, the records of a slice being added up on a second stream while
    the next slice sweeps into the other buffer set (capi_solvers.hip: bahip_optimize_intrinsics).  Forced on the small scene: the
    same bits as the oracle, whatever the number of slices, also with buffers that overflow into the direct path."""
    import ctypes as C
    for capacity in (-1, 64):
        ba, g = _perturbed_pair(scene)
        ba.use_depth, ba.use_desc = 1, 1
        lib, h = g.ctx.lib, g.ctx.handle
        assert lib.bahip_debug_set_intrinsics_slices(h, slices) == 0
        assert lib.bahip_debug_set_intrinsics_bin_capacity(h, capacity) == 0
        for _ in range(2):                       # the second call reuses the buffer sets and the second stream
            _, dc_r, a_r = ba.optimize_intrinsics(True, True)
            _, dc_g, a_g = g.optimize_intrinsics(True, True)
            assert np.array_equal(_bits(_cam_tuple(dc_g)), _bits(_cam_tuple(dc_r))), (slices, capacity)
            assert np.array_equal(_bits([a_g]), _bits([a_r]))
            assert np.array_equal(_bits(g.cfactor.download()), _bits(ba.cfactor))
        cap, most, total = C.c_uint32(), C.c_uint32(), C.c_uint64()
        assert lib.bahip_debug_intrinsics_bin_stats(h, C.byref(cap), C.byref(most), C.byref(total)) == 0
        assert total.value > 100000                # the pairs of all slices went through the reservation


def test_color_intrinsics_step(scene):
    ba, g = _perturbed_pair(scene, depth_cam_offset=(0, 0, 0, 0), color_cam_offset=(0.4, -0.3, 0.8, -0.6))
    ba.use_depth, ba.use_desc = 1, 1
    cc_r, _, _ = ba.optimize_intrinsics(False, True)
    cc_g, _, _ = g.optimize_intrinsics(False, True)
    assert np.array_equal(_bits(_cam_tuple(cc_g)), _bits(_cam_tuple(cc_r))), (_cam_tuple(cc_g), _cam_tuple(cc_r))


def _pcg_setup(scene, mode):
    rng = np.random.Generator(np.random.PCG64(33))
    ba, g = _perturbed_pair(scene, depth_cam_offset=(0, 0, 0, 0) if mode != "all" else (0.3, -0.2, 0.5, -0.4))
    data, _ = common.oracle_surfels(ba)
    data[2] += rng.uniform(0, 0.003, data.shape[1]).astype(np.float32)
    ba.surfel_data[:, :data.shape[1]] = data
    g.upload_surfels(data, np.ones(data.shape[1], np.uint8))
    perturbed = [common.synthetic.perturb_pose(rng, T, 0.003, 0.0005) for T in scene.poses_gt]
    for k, T in enumerate(perturbed):
        ba.set_pose(k, T)
        g.keyframes[k]["pose"] = np.asarray(T, np.float32)
    g.bind_keyframes()
    ba.use_depth, ba.use_desc = 1, 1
    # With increase_ba_iteration_count == false the reference runs the end-of-scheme tasks (surfel
    # deletion, compaction) first unless they already ran for this iteration count; skip them here.
    ba.last_ba_iteration_count = ba.ba_iteration_count
    return ba, g, data, perturbed


@pytest.mark.parametrize("mode", ["poses+geometry", "all"])
def test_pcg_system_assembly(scene, mode):
    """r = -J^T W F and M = diag(J^T W J) (PCGInit over all keyframes): every entry, bit for bit."""
    ba, g, data, _ = _pcg_setup(scene, mode)
    di = ci = (mode == "all")
    r_ref, M_ref = ba.pcg_assemble(True, True, di, ci, gauge_keyframe=1)
    g.pcg_iteration(optimize_poses=True, optimize_geometry=True, optimize_depth_intrinsics=di, optimize_color_intrinsics=ci,
                    max_inner_iterations=0, gauge_keyframe=1)
    U = len(r_ref)
    r, M = g.read_pcg_vector(0, U), g.read_pcg_vector(1, U)
    K, N = len(ba.keyframes), data.shape[1]
    ps = 6 * (K - 1)
    assert U == ps + 3 * N + ((5 + ba.cf_w * ba.cf_h + 4) if di else 0)
    assert np.count_nonzero(M_ref[:ps]) == ps and np.count_nonzero(r_ref[:ps]) == ps
    if di:
        assert np.count_nonzero(M_ref[ps + 3 * N:]) > 0.5 * (U - ps - 3 * N)
    # surfel block: per-surfel chains over the keyframes; dense head (poses, intrinsics, cfactor cells): exact sums of tile /
    # pair terms -- the same bits although the kernel adds them with atomics in arbitrary order
    assert np.array_equal(_bits(r), _bits(r_ref)), np.flatnonzero(_bits(r) != _bits(r_ref))[:10]
    assert np.array_equal(_bits(M), _bits(M_ref)), np.flatnonzero(_bits(M) != _bits(M_ref))[:10]
    # and run to run (the reference's PCG is not reproducible: float atomics)
    g.pcg_iteration(optimize_poses=True, optimize_geometry=True, optimize_depth_intrinsics=di, optimize_color_intrinsics=ci,
                    max_inner_iterations=0, gauge_keyframe=1)
    assert np.array_equal(_bits(g.read_pcg_vector(0, U)), _bits(r))


@pytest.mark.parametrize("mode,lds_form", [("poses+geometry", 0), ("geometry-only", 0), ("all", 0), ("poses+geometry", 2), ("all", 2)])
def test_pcg_iteration(scene, mode, lds_form, request):
    """One outer iteration of the PCG scheme: the same number of inner steps, the same surfels, poses and calibration, bit for
    bit (with a != 0 in the joint mode: exp_det).  lds_form 2: the step-1 sweep by persistent workgroups that keep the pose block
    of the dense head in LDS (round 4; the form the bench size takes) -- exact integer sums, the same bits."""
    from badslam_amd import capi as _capi
    _capi.check(_capi.load().bahip_debug_set_pcg_lds_form(lds_form))
    request.addfinalizer(lambda: _capi.check(_capi.load().bahip_debug_set_pcg_lds_form(1)))
    ba, g, data, perturbed = _pcg_setup(scene, mode)
    poses_on = mode != "geometry-only"
    di = ci = (mode == "all")
    if mode == "all":        # a != 0: the (defined) exponential of the depth deformation is in every depth Jacobian
        ba.dp.a = 0.0125
        g.dp.a = 0.0125
        g.set_intrinsics()
    cost_before, _ = ba.evaluate_cost()
    stats = ba.bundle_adjustment(optimize_depth_intrinsics=di, optimize_color_intrinsics=ci, optimize_poses=poses_on,
                                 optimize_geometry=True, min_iterations=1, max_iterations=1, use_pcg=True,
                                 increase_ba_iteration_count=False, pcg_gauge_keyframe=0)
    g.active_buf.upload(np.ones((1, g.capacity), np.uint8))
    g.update_surfel_normals()
    steps, conv = g.pcg_iteration(optimize_poses=poses_on, optimize_geometry=True, optimize_depth_intrinsics=di,
                                  optimize_color_intrinsics=ci, gauge_keyframe=0)
    assert 3 <= stats.pcg_inner_steps_total <= 30
    assert steps == stats.pcg_inner_steps_total, (steps, stats.pcg_inner_steps_total)
    got = g.download_surfels()
    ref = ba.surfel_data[:, :got.shape[1]].copy()
    moved = np.abs(ref[:3] - data[:3]).max(axis=0)
    assert np.median(moved) > 1e-4                                                  # a real update happened
    cost_ref, _ = ba.evaluate_cost()
    assert cost_ref < 0.7 * cost_before
    assert np.array_equal(got[:8].view(np.uint32), ref[:8].view(np.uint32)), np.abs(got[:3] - ref[:3]).max()
    for k in range(len(perturbed)):
        assert np.array_equal(np.asarray(g.keyframes[k]["pose"], np.float32), np.asarray(ba.pose(k), np.float32)), k
    assert np.array_equal(_bits(_cam_tuple(g.depth_cam)), _bits(_cam_tuple(ba.depth_cam)))
    assert np.array_equal(_bits(_cam_tuple(g.color_cam)), _bits(_cam_tuple(ba.color_cam)))
    assert np.array_equal(_bits([g.dp.a]), _bits([ba.dp.a]))
    assert np.array_equal(_bits(g.cfactor.download()), _bits(ba.cfactor))
