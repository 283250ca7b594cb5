"""Oracle parity at the scale of BASELINE.json's configs ("next round" item 1).

(a) K = 200 keyframes (160x120): the first place where the wave-level keyframe culling (wave_cull.h: for_each_candidate)
    walks several 64-keyframe chunks and splits them over `parts` wavefronts against a checker: creation, activation
    flags and the geometry step bit for bit (both launch shapes), batched pose estimation with pose_parts 1 / 2 / 8.
(b) One full iteration of BASELINE configs[2] (200 keyframes x 3 M surfels x 640x480, bench.build_scene) through the
    oracle's BundleAdjustment restatement and through vis::DirectBA: surfels bit for bit, poses <= 1e-5 m RMSE.
(c) Sampled per-pair parity at that size: bahip_debug_evaluate_pairs vs orc_evaluate_pairs on 10^6 random
    (surfel, keyframe) pairs -- association decision, pixel, calibrated depth, the three residuals, weights, pose
    Jacobians and image gradients, every bit.
(d) BASELINE configs[1]'s size (50 keyframes, ~500 k surfels, 640x480; synthetic stand-in for TUM fr1/desk) end to end:
    alternating BA with the surfel lifecycle inside the loop (filtered creation with the host's co-visibility lists,
    merge, delete + radii, compaction) against the oracle.
The oracle gets the very images the HIP path works on (downloaded preprocessed keyframes); preprocessing parity has its
own tests (test_gpu_kernels_vs_oracle.py)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from badslam_amd import capi, synthetic
from tests import common

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("BENCH_KEEP_FRAMES", "1")   # bench.build_scene keeps the rendered frames: test_c3_eight_surfel_shards_... builds the ranks' instances from them


def _shapes(lib, tile_waves, pose_parts):
    capi.check(lib.bahip_debug_set_launch_shapes(tile_waves, pose_parts))


# ---- (a) many keyframes ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def many():
    scene = synthetic.make_scene(200, 160, 120, seed=21, cell=2, translation_range=5.0, rotation_range=0.9)
    orc = common.build_oracle(scene, 900000)
    g = common.build_gpu(scene, 900000)           # the GPU creates its own surfels from all 200 keyframes
    return scene, orc, g


def test_many_keyframes_creation_bit_exact(many):
    scene, orc, g = many
    ref, _ = common.oracle_surfels(orc)
    got = g.download_surfels()
    assert got.shape[1] == ref.shape[1] > 100000, (got.shape, ref.shape)
    assert np.array_equal(got[:8].view(np.uint32), ref[:8].view(np.uint32))


def _set_activations(orc, g, pattern):
    for k in range(len(orc.keyframes)):
        orc.keyframes[k].activation = int(pattern[k])
        g.keyframes[k]["activation"] = int(pattern[k])


@pytest.mark.parametrize("tile_waves,fused,classes", [(1, False, 4), (4, False, 4), (1, True, 4), (4, True, 4), (1, True, 8), (4, True, 8), (4, False, 8)])
def test_many_keyframes_activation_and_geometry_bit_exact(many, tile_waves, fused, classes, request):
    """classes: the number of interleaved keyframe classes the per-surfel sums are defined over -- 4 by default, 8 for keyframe
    sharding over 8 ranks (bahip_context_set_sum_classes / orc_set_sum_classes): either definition, both launch shapes, the
    oracle's bits."""
    from oracle import binding as ob
    scene, orc, g = many
    K = len(orc.keyframes)
    _shapes(g.ctx.lib, tile_waves, 0)
    g.set_sum_classes(classes)
    ob.lib().orc_set_sum_classes(classes)
    request.addfinalizer(lambda: (_shapes(g.ctx.lib, 0, 0), g.set_sum_classes(4), ob.lib().orc_set_sum_classes(4)))
    orc.use_depth, orc.use_desc = 1, 1
    data, active = common.oracle_surfels(orc)
    n = data.shape[1]
    rng = np.random.Generator(np.random.PCG64(31))
    moved = data.copy()
    moved[2] += rng.uniform(0, 0.004, n).astype(np.float32)
    moved[6] += 2.0
    # a third of the keyframes active, a third co-visible active, a third inactive: activation looks at the active ones
    # only, the geometry passes skip the inactive ones (predicates of wave_cull.h candidates differ per pass)
    _set_activations(orc, g, np.arange(K) % 3)
    try:
        orc.surfel_data[:, :n] = moved
        g.upload_surfels(moved, active * 0)
        g.bind_keyframes()
        orc.update_surfel_activation()
        if fused:          # activation decided inside the normals pass of the geometry sweep (one launch instead of two)
            g.update_activation_and_optimize_geometry(True, True)
        else:
            g.update_surfel_activation()
        act = g.active_buf.download()[0, :n]
        assert np.array_equal(act, orc.active[:n])
        assert 0.3 * n < act.sum() < n                 # some surfels are seen by no active keyframe
        if not fused:
            g.optimize_geometry_iteration(True, True)
        orc.optimize_geometry_iteration()
        got = g.download_surfels()
        ref = orc.surfel_data[:, :n]
        assert np.array_equal(got[:8].view(np.uint32), ref[:8].view(np.uint32))
        assert np.abs(ref[2] - moved[2])[act == 1].mean() > 1e-4
    finally:
        _set_activations(orc, g, np.zeros(K, int))
        orc.surfel_data[:, :n] = data


_ORACLE_POSES = {}


def _oracle_pose_estimates(orc, perturbed, pattern):
    """orc_estimate_frame_pose of every non-inactive keyframe (independent of each other; ctypes releases the GIL)."""
    if not _ORACLE_POSES:
        from concurrent.futures import ThreadPoolExecutor
        ks = [k for k in range(len(perturbed)) if pattern[k] != capi.KF_INACTIVE]
        with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 8)) as pool:
            for k, res in zip(ks, pool.map(lambda k: orc.estimate_frame_pose(k, perturbed[k]), ks)):
                _ORACLE_POSES[k] = res
    return _ORACLE_POSES


# The persistent LDS form as a regression matrix: wavefronts per workgroup 1 / 2 / 16 x work items per
# launch 1 / 37 / 292 (200 keyframes: 200 slices, 6 slices, one launch), then the shapes round 4 added: a tile's work items
# shared by 4 wavefronts (p2), and Gauss-Newton rounds queued ahead of the host one / four at a time (a1 / a4).
_LDS_MATRIX = [f"lds:w{w}:i{i}" for w in (1, 2, 16) for i in (1, 37, 292)] + ["lds:w16:i292:p2", "lds:w2:i292:p3", "lds:w16:i292:a1", "lds:w16:i292:a4",
                                                                               "lds:w16:i292:p1:a4"]


@pytest.mark.parametrize("pose_parts", [1, 2, 8, "lds", "lds-sliced", "global:a4"] + _LDS_MATRIX)
def test_many_keyframes_batched_pose_estimation(many, pose_parts, request):
    """pose_parts 1 / 2 / 8: one tile per wavefront, a tile's work items split over that many wavefronts, totals added to the
    normal equations with global integer atomics; "lds": persistent workgroups that keep the normal equations of all 200 work
    items in LDS and flush once (the form the bench size takes); "lds-sliced": the same with the work items cut into slices of
    48 per launch (what happens beyond 292 work items, e.g. the 1000 keyframes of configs[4]; later rounds with fewer items
    left take a single launch).  The sums are integer sums: the same bits for every form."""
    scene, orc, g = many
    K = len(orc.keyframes)
    lib = g.ctx.lib
    if isinstance(pose_parts, str) and ":" in pose_parts:
        form, *options = pose_parts.split(":")
        opt = {o[0]: int(o[1:]) for o in options}
        capi.check(lib.bahip_debug_set_pose_form(2 if form == "lds" else 1))
        capi.check(lib.bahip_debug_set_pose_lds_items(opt.get("i", 0) if opt.get("i", 292) < 292 else 0))
        capi.check(lib.bahip_debug_set_pose_lds_shape(opt.get("w", 0), opt.get("p", -1)))
        capi.check(lib.bahip_debug_set_pose_rounds_ahead(opt.get("a", 0)))
        request.addfinalizer(lambda: (capi.check(lib.bahip_debug_set_pose_form(0)), capi.check(lib.bahip_debug_set_pose_lds_items(0)),
                                      capi.check(lib.bahip_debug_set_pose_lds_shape(0, -1)), capi.check(lib.bahip_debug_set_pose_rounds_ahead(0))))
    elif pose_parts in ("lds", "lds-sliced"):
        capi.check(g.ctx.lib.bahip_debug_set_pose_form(2))
        capi.check(g.ctx.lib.bahip_debug_set_pose_lds_items(48 if pose_parts == "lds-sliced" else 0))
        request.addfinalizer(lambda: (capi.check(g.ctx.lib.bahip_debug_set_pose_form(0)), capi.check(g.ctx.lib.bahip_debug_set_pose_lds_items(0))))
    else:
        capi.check(g.ctx.lib.bahip_debug_set_pose_form(1))
        _shapes(g.ctx.lib, 0, pose_parts)
        request.addfinalizer(lambda: (_shapes(g.ctx.lib, 0, 0), capi.check(g.ctx.lib.bahip_debug_set_pose_form(0))))
    orc.use_depth, orc.use_desc = 1, 1
    data, active = common.oracle_surfels(orc)
    g.upload_surfels(data, active)
    rng = np.random.Generator(np.random.PCG64(41))
    perturbed = [synthetic.perturb_pose(rng, T) for T in scene.poses_gt]
    pattern = np.where(np.arange(K) % 7 == 3, capi.KF_INACTIVE, np.arange(K) % 2)     # inactive keyframes are skipped
    _set_activations(orc, g, pattern)
    try:
        for k, T in enumerate(perturbed):
            orc.set_pose(k, T)
            g.keyframes[k]["pose"] = np.asarray(T, np.float32)
        g.bind_keyframes()
        poses, its, conv, rounds = g.estimate_keyframe_poses(True, True)
        reference = _oracle_pose_estimates(orc, perturbed, pattern)
        for k in range(K):
            if pattern[k] == capi.KF_INACTIVE:
                assert its[k] == 0
                assert np.array_equal(poses[k].astype(np.float32), np.asarray(perturbed[k], np.float32))
                continue
            est, its_ref, conv_ref = reference[k]
            # The normal equations are a defined sum (per-tile tree, fixed-point total: ba_device.h HbFixed, oracle_pose.c), the
            # 6x6 solve is binary64 on both sides and the exponential map uses the same defined sin / cos: every Gauss-Newton
            # step -- and so the number of steps and the final pose -- is the same bits, whatever pose_parts is.
            assert its[k] == its_ref and conv[k] == int(conv_ref), (k, its[k], its_ref)
            assert np.array_equal(poses[k].astype(np.float32), est.to_array().astype(np.float32)), (k, common.pose_error(est.to_array(), poses[k]))
        assert rounds == its.max() >= 2
    finally:
        _set_activations(orc, g, np.zeros(K, int))
        for k, T in enumerate(scene.poses_gt):
            orc.set_pose(k, T)
            g.keyframes[k]["pose"] = np.asarray(T, np.float32)


def test_many_keyframes_intrinsics_step_bit_exact(many):
    """The intrinsics sweep (kernels_intrinsics.hip) over 200 keyframes -- several 64-keyframe chunks in its candidate loop --
    with perturbed cameras: depth camera, deformation parameter, every cfactor cell and the colour camera come out with the
    oracle's bits (the sums are defined: DESIGN.md section 3)."""
    scene, orc, g = many
    K = len(orc.keyframes)
    orc.use_depth, orc.use_desc = 1, 1
    data, active = common.oracle_surfels(orc)
    g.upload_surfels(data, np.ones_like(active))
    orc.active[:data.shape[1]] = 1
    _set_activations(orc, g, np.zeros(K, np.int64))
    for k, T in enumerate(scene.poses_gt):
        orc.set_pose(k, T)
        g.keyframes[k]["pose"] = np.asarray(T, np.float32)
    saved = [(obj, name, tuple(getattr(getattr(obj, name), f) for f in ("fx", "fy", "cx", "cy"))) for obj in (orc, g)
             for name in ("depth_cam", "color_cam")]
    cf_saved, a_saved = orc.cfactor.copy(), orc.dp.a
    try:
        for obj in (orc, g):
            obj.depth_cam.fx += 0.5; obj.depth_cam.fy -= 0.6; obj.depth_cam.cx += 1.23; obj.depth_cam.cy -= 1.17
            obj.color_cam.fx += 0.4; obj.color_cam.fy -= 0.3; obj.color_cam.cx += 0.8; obj.color_cam.cy -= 0.6
        g.set_intrinsics()
        g.bind_keyframes()
        cc_r, dc_r, a_r = orc.optimize_intrinsics(True, True, apply=False)
        assert np.count_nonzero(orc.cfactor) > 0.5 * orc.cfactor.size
        # both forms of the record reduction (kernels_intrinsics.hip), each once with the record buffers sized by the call before
        # (the second call of a form: every record goes through the buffers, several chunks and slices per buffer at this size)
        lib = g.ctx.lib
        try:
            for form in (0, 0, 1, 1):
                assert lib.bahip_debug_set_intrinsics_reduce_form(form) == 0
                g.cfactor.upload(cf_saved)
                cc_g, dc_g, a_g = g.optimize_intrinsics(True, True, apply=False)
                for got, ref in ((dc_g, dc_r), (cc_g, cc_r)):
                    assert (got.fx, got.fy, got.cx, got.cy) == (ref.fx, ref.fy, ref.cx, ref.cy), (form, (got.fx, got.fy, got.cx, got.cy), (ref.fx, ref.fy, ref.cx, ref.cy))
                assert np.float32(a_g) == np.float32(a_r), form
                cf_g = g.cfactor.download()
                assert np.array_equal(cf_g.view(np.uint32), orc.cfactor.view(np.uint32)), form
        finally:
            assert lib.bahip_debug_set_intrinsics_reduce_form(-1) == 0
    finally:
        for obj, name, values in saved:
            cam = getattr(obj, name)
            cam.fx, cam.fy, cam.cx, cam.cy = values
        orc.cfactor[:] = cf_saved
        orc.dp.a = a_saved
        g.dp.a = a_saved
        g.cfactor.upload(cf_saved)
        g.set_intrinsics()


def test_many_keyframes_pcg_assembly(many):
    """PCGInit over 200 keyframes: the surfel block of r = -J^T W F and M = diag(J^T W J) bit for bit (per-surfel sums in
    keyframe order) and the 6 x 199 pose block (exact sums): bit for bit."""
    scene, orc, g = many
    K = len(orc.keyframes)
    orc.use_depth, orc.use_desc = 1, 1
    data, active = common.oracle_surfels(orc)
    g.upload_surfels(data, np.ones_like(active))
    orc.active[:data.shape[1]] = 1
    _set_activations(orc, g, np.zeros(K, np.int64))
    rng = np.random.Generator(np.random.PCG64(43))
    perturbed = [synthetic.perturb_pose(rng, T) for T in scene.poses_gt]
    for k, T in enumerate(perturbed):
        orc.set_pose(k, T)
        g.keyframes[k]["pose"] = np.asarray(T, np.float32)
    g.bind_keyframes()
    g.update_surfel_normals()
    orc.update_surfel_normals()
    r_ref, M_ref = orc.pcg_assemble(True, True, False, False, gauge_keyframe=1)
    g.pcg_iteration(optimize_poses=True, optimize_geometry=True, optimize_depth_intrinsics=False, optimize_color_intrinsics=False,
                    max_inner_iterations=0, gauge_keyframe=1)
    U = len(r_ref)
    r, M = g.read_pcg_vector(0, U), g.read_pcg_vector(1, U)
    N = data.shape[1]
    ps = 6 * (K - 1)
    assert U == ps + 3 * N
    # every entry, the 6 x 199 pose block (exact sums of ~1e3 tile totals per entry, added with atomics in arbitrary order by
    # the kernel) included
    assert np.array_equal(r.view(np.uint32), r_ref.view(np.uint32)), np.flatnonzero(r.view(np.uint32) != r_ref.view(np.uint32))[:10]
    assert np.array_equal(M.view(np.uint32), M_ref.view(np.uint32)), np.flatnonzero(M.view(np.uint32) != M_ref.view(np.uint32))[:10]
    assert np.count_nonzero(M_ref[:ps]) > 0.8 * ps       # (a few keyframes of this scene see no surfel)
    for k, T in enumerate(scene.poses_gt):     # leave the shared fixture as it was
        orc.set_pose(k, T)
        g.keyframes[k]["pose"] = np.asarray(T, np.float32)


# ---- (b), (c): BASELINE configs[2] at full size -------------------------------------------------------------------------
def _bench_scene(**overrides):
    sys.path.insert(0, ROOT)
    import bench
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        args = bench.parse_args()
    finally:
        sys.argv = argv
    for k, v in overrides.items():
        setattr(args, k, v)
    ba, data, poses_gt = bench.build_scene(args, lambda m: None)
    args.frames = bench.build_scene.frames   # (the rendered frames of THIS scene: the next call of build_scene replaces the attribute)
    return ba, data, poses_gt, args


def _oracle_from_directba(ba, args, data, keyframe_ids=None):
    """Oracle scene holding the preprocessed images of `keyframe_ids` (default: all) and the surfel rows `data`."""
    cam = synthetic.test_camera(args.width, args.height)
    from oracle import binding as ob
    orc = ob.OracleBA(data.shape[1] + 1024, 1.0 / 5000, 40.0, args.cell, ob.make_camera(cam, args.width, args.height),
                      ob.make_camera(cam, args.width, args.height))
    ids = range(ba.keyframe_count()) if keyframe_ids is None else keyframe_ids
    for k in ids:
        orc.add_preprocessed_keyframe(ba.keyframe_image(k, "depth"), ba.keyframe_image(k, "normals"), ba.keyframe_image(k, "radius"),
                                      ba.keyframe_image(k, "color"), ba.keyframe_pose(k))
    n = data.shape[1]
    orc.surfel_data[:data.shape[0], :n] = data
    orc.surfels.surfels_size = orc.surfels.surfel_count = n
    return orc


@pytest.fixture(scope="module")
def c3():
    ba, data, poses_gt, args = _bench_scene()
    ba.upload_surfels(data)
    return ba, data, poses_gt, args


def test_c3_sampled_pairs_bit_exact(c3):
    ba, data, poses_gt, args = c3
    ba.upload_surfels(data)
    N, K = data.shape[1], ba.keyframe_count()
    assert (N, K) == (3000000, 200)
    rng = np.random.Generator(np.random.PCG64(77))
    kf_ids = sorted(rng.choice(K, 25, replace=False).tolist())
    orc = _oracle_from_directba(ba, args, data, kf_ids)
    ctx = ba.backend_context()
    ba.BindScene()                                     # cameras + depth parameters into the backend context
    surfels = ba.surfels_struct()
    F_PAIR = orc.PAIR_FIELDS
    total = associated = with_colour = 0
    for j, k in enumerate(kf_ids):
        idx = rng.integers(0, N, 40000).astype(np.uint32)
        F = np.array(list(orc.keyframes[j].frame_T_global), np.float32)
        ref = orc.evaluate_pairs(j, idx)
        out = np.zeros((len(idx), 40), np.float32)
        frame = ba.keyframe_frame(k)
        capi.check(ctx.lib.bahip_debug_evaluate_pairs(ctx.handle, C.byref(frame), F.ctypes.data_as(C.POINTER(C.c_float)),
                                                      C.byref(surfels), idx.ctypes.data_as(C.POINTER(C.c_uint32)), len(idx),
                                                      out.ctypes.data_as(C.POINTER(C.c_float))))
        got = out.view(np.uint32)
        refi = ref.view(np.int32)
        assoc = refi[:, 0] == 1
        assert np.array_equal(out[:, 0] == 1.0, assoc), k                              # the association decision of every pair
        a = np.flatnonzero(assoc)
        assert np.array_equal(out[a, 1].astype(np.int32), refi[a, 1]) and np.array_equal(out[a, 2].astype(np.int32), refi[a, 2])
        assert np.array_equal(out[a, 3] == 1.0, refi[a, 3] == 1)

        def same(gpu_cols, field, rows):
            o, n = F_PAIR[field]
            x, y = got[rows][:, gpu_cols].copy(), ref[rows][:, o:o + n].copy()
            x[x == 0x80000000] = 0           # -0.0 and +0.0 are the same value (a compiler may turn -fma(a, b, -c) into
            y[y == 0x80000000] = 0           # fma(-a, b, c), which differs in the sign of an exact zero only)
            if not np.array_equal(x, y):
                bad = np.argwhere(x != y)
                detail = [(int(idx[rows[r]]), int(c), float(x[r, c:c + 1].view(np.float32)[0]), float(y[r, c:c + 1].view(np.float32)[0]))
                          for r, c in bad[:8]]
                raise AssertionError(f"keyframe {k}, {field}: {len(bad)} words differ in {len(set(bad[:, 0]))} of {len(rows)} pairs; "
                                     f"(surfel, column, hip, oracle): {detail}")

        same([4], "calibrated_depth", a)
        same([5], "depth_residual", a)
        same([6], "depth_weight", a)
        same([7], "depth_inv_stddev", a)
        same(list(range(8, 14)), "depth_jac_pose", a)
        c = a[refi[a, 3] == 1]
        same([14, 15], "desc_residual", c)
        same([16, 17], "desc_weight", c)
        same(list(range(30, 34)), "grad", c)
        same(list(range(18, 30)), "desc_jac_pose", c)
        total += len(idx); associated += len(a); with_colour += len(c)
    assert total == 1000000
    assert associated > 50000 and with_colour > 0.9 * associated, (associated, with_colour)
    print(f"{total} pairs, {associated} associated, {with_colour} with a colour pixel: every compared field bit-identical")


def test_c3_one_iteration_matches_oracle(c3):
    ba, data, poses_gt, args = c3
    K = ba.keyframe_count()
    ba.upload_surfels(data)
    start = [ba.keyframe_pose(k) for k in range(K)]
    orc = _oracle_from_directba(ba, args, data)
    try:
        ba.set_ba_iteration_counts(1, 1)               # equal counters: no end-of-scheme tasks (fixed surfel set)
        done, _ = ba.BundleAdjustment(do_surfel_updates=False, optimize_poses=True, optimize_geometry=True, min_iterations=1,
                                      max_iterations=1, active_keyframe_window_start=0, active_keyframe_window_end=K - 1,
                                      increase_ba_iteration_count=False)
        orc.ba_iteration_count, orc.last_ba_iteration_count = 1, 1
        stats = orc.bundle_adjustment(do_surfel_updates=False, optimize_poses=True, optimize_geometry=True, min_iterations=1,
                                      max_iterations=1, increase_ba_iteration_count=False)
        assert done == stats.iterations_done == 1
        n = data.shape[1]
        got = ba.download_surfels(8)
        ref = orc.surfel_data[:8, :n]
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))              # geometry step at full size: every bit
        assert np.abs(ref[2] - data[2]).mean() > 1e-4
        got_poses = np.array([ba.keyframe_pose(k) for k in range(K)])
        ref_poses = np.array([orc.pose(k) for k in range(K)])
        moved = np.linalg.norm(got_poses[:, 4:] - np.array(start)[:, 4:], axis=1)
        assert np.median(moved) > 1e-3                                               # the 5 mm perturbation was worked on
        rmse = float(np.sqrt(np.mean(np.sum((got_poses[:, 4:] - ref_poses[:, 4:]) ** 2, axis=1))))
        identical = int(np.sum(np.all(got_poses.astype(np.float32) == ref_poses.astype(np.float32), axis=1)))
        print(f"config 3, one iteration: pose RMSE vs oracle {rmse:.2e} m, {identical} of {K} poses bit-identical, "
              f"{stats.pose_gn_steps_total} GN steps in the oracle")
        assert rmse <= 1e-5, rmse                                                    # the north-star gate
        assert identical == K                                                        # and in fact every bit
        assert ba.last_stats()["pose_steps"] == stats.pose_gn_steps_total
    finally:
        for k in range(K):
            ba.set_keyframe_pose(k, start[k])


def test_c3_fast_flavour_against_the_exact_build(c3):
    """the fast arithmetic flavour (bahip_context_set_arithmetic) at BASELINE configs[2] against the bit-exact
    build -- three alternating iterations from the perturbed start: pose RMSE <= 1e-5 m (BASELINE.json's bar; the exact build is the
    oracle's bits, test_c3_one_iteration_matches_oracle), surfel positions alike, and the association decisions of 10^6 sampled
    (surfel, keyframe) pairs compared one by one: flips <= 0.1 %."""
    ba, data, poses_gt, args = c3
    K, N = ba.keyframe_count(), data.shape[1]
    start = [ba.keyframe_pose(k) for k in range(K)]
    ctx = ba.backend_context()
    results = {}
    try:
        for name, fast in (("exact", False), ("fast", True)):
            ba.SetFastArithmetic(fast)
            ba.upload_surfels(data)
            for k in range(K):
                ba.set_keyframe_pose(k, start[k])
            ba.set_ba_iteration_counts(1, 1)
            done, _ = ba.BundleAdjustment(do_surfel_updates=False, optimize_poses=True, optimize_geometry=True, min_iterations=3,
                                          max_iterations=3, active_keyframe_window_start=0, active_keyframe_window_end=K - 1,
                                          increase_ba_iteration_count=False)
            assert done == 3
            results[name] = (np.array([ba.keyframe_pose(k) for k in range(K)]), ba.download_surfels(8))
        # association decisions pair by pair, both flavours on the SAME state (the fast run's result)
        ba.BindScene()
        surfels = ba.surfels_struct()
        from badslam_amd import se3
        kf_ids = sorted(np.random.Generator(np.random.PCG64(78)).choice(K, 25, replace=False).tolist())
        decisions = {}
        for name, fast in (("exact", False), ("fast", True)):
            ba.SetFastArithmetic(fast)
            pair_rng = np.random.Generator(np.random.PCG64(79))         # the same pairs for both flavours
            outs = []
            for k in kf_ids:
                idx = pair_rng.integers(0, N, 40000).astype(np.uint32)
                F = np.ascontiguousarray(se3.matrix(se3.inverse(ba.keyframe_pose(k)))[:3, :4], np.float32).reshape(-1)
                out = np.zeros((len(idx), 40), np.float32)
                frame = ba.keyframe_frame(k)
                capi.check(ctx.lib.bahip_debug_evaluate_pairs(ctx.handle, C.byref(frame), F.ctypes.data_as(C.POINTER(C.c_float)),
                                                              C.byref(surfels), idx.ctypes.data_as(C.POINTER(C.c_uint32)), len(idx),
                                                              out.ctypes.data_as(C.POINTER(C.c_float))))
                outs.append(out[:, [0, 4, 5, 14, 15]].copy())
            decisions[name] = np.concatenate(outs)
    finally:
        ba.SetFastArithmetic(False)
        for k in range(K):
            ba.set_keyframe_pose(k, start[k])
    (pe, re_), (pf, rf) = results["exact"], results["fast"]
    rmse = float(np.sqrt(np.mean(np.sum((pe[:, 4:] - pf[:, 4:]) ** 2, axis=1))))
    worst = float(np.max(np.linalg.norm(pe[:, 4:] - pf[:, 4:], axis=1)))
    moved = np.linalg.norm(pe[:, 4:] - np.array(start)[:, 4:], axis=1)
    dpos = np.abs(re_[:3] - rf[:3]).max(axis=0)
    far = int(np.count_nonzero(~(dpos <= 1e-5)))
    de, df = decisions["exact"], decisions["fast"]
    pairs, associated = len(de), int(np.count_nonzero(de[:, 0] == 1.0))
    flips = int(np.count_nonzero(de[:, 0] != df[:, 0]))
    both = (de[:, 0] == 1.0) & (df[:, 0] == 1.0)
    ddepth = np.abs(de[both, 1] - df[both, 1]).max()
    dres = np.abs(de[both, 2] - df[both, 2])
    print(f"config 3, fast vs exact flavour after 3 iterations: pose RMSE {rmse:.2e} m (max {worst:.2e} m; the poses moved {np.median(moved):.1e} m), "
          f"{far} of {N} surfels beyond 1e-5 m (p99.9 {np.percentile(dpos, 99.9):.1e} m); {pairs} sampled pairs, {associated} associated, "
          f"{flips} association flips ({flips / max(associated, 1):.2e} of the associated), calibrated depth within {ddepth:.1e} m, "
          f"depth residual within {dres.max():.1e} (median {np.median(dres):.1e})")
    assert np.median(moved) > 1e-3
    assert rmse <= 1e-5 and worst <= 1e-5, (rmse, worst)
    assert far <= 1e-3 * N, far
    assert pairs == 1000000 and associated > 50000
    assert flips <= 1e-3 * associated, (flips, associated)
    assert not np.array_equal(pe.astype(np.float32), pf.astype(np.float32)), "the fast flavour gave the exact flavour's bits: was it selected?"


# ---- (d): BASELINE configs[1]'s size, end to end with the surfel lifecycle ------------------------------------------------
def test_c2_size_end_to_end_with_surfel_updates():
    ba, data, poses_gt, args = _bench_scene(keyframes=50, surfels=10 ** 9, no_spatial_sort=True)
    K = ba.keyframe_count()
    # start from an empty cloud: BundleAdjustment creates the surfels itself (filtered, co-visibility lists of the host)
    ba.SetSurfelCount(0, 0)
    cam = synthetic.test_camera(args.width, args.height)
    from oracle import binding as ob
    cap = data.shape[1] * 2 + 1024
    orc = ob.OracleBA(cap, 1.0 / 5000, 40.0, args.cell, ob.make_camera(cam, args.width, args.height),
                      ob.make_camera(cam, args.width, args.height), min_observation_count=2)
    for k in range(K):
        orc.add_preprocessed_keyframe(ba.keyframe_image(k, "depth"), ba.keyframe_image(k, "normals"), ba.keyframe_image(k, "radius"),
                                      ba.keyframe_image(k, "color"), ba.keyframe_pose(k))
    orc.covis = [ba.keyframe_covisibility(k) for k in range(K)]
    assert 1 <= min(len(l) for l in orc.covis) < K - 1                          # a real co-visibility structure, not "all with all"
    orc.spatial_sort_cell, orc.unsorted_surfels = 0.02, ba.unsorted_surfels()   # the end tasks put the buffer in Morton order on both sides
    for call in range(2):
        done, _ = ba.BundleAdjustment(do_surfel_updates=True, optimize_poses=True, optimize_geometry=True, min_iterations=2,
                                      max_iterations=2, increase_ba_iteration_count=True)
        stats = orc.bundle_adjustment(do_surfel_updates=True, optimize_poses=True, optimize_geometry=True, min_iterations=2,
                                      max_iterations=2, increase_ba_iteration_count=True)
        assert done == stats.iterations_done == 2
        assert ba.surfel_count() == orc.surfels_size, (call, ba.surfel_count(), orc.surfels_size)
    n = orc.surfels_size
    assert 300000 < n < 900000, n
    got_poses = np.array([ba.keyframe_pose(k) for k in range(K)])
    ref_poses = np.array([orc.pose(k) for k in range(K)])
    rmse = float(np.sqrt(np.mean(np.sum((got_poses[:, 4:] - ref_poses[:, 4:]) ** 2, axis=1))))
    got = ba.download_surfels(8)
    ref = orc.surfel_data[:8, :n]
    dpos = np.abs(got[:3] - ref[:3]).max(axis=0)
    flips = int(np.count_nonzero(~(dpos <= 1e-5)))
    print(f"config-2 size: {n} surfels, pose RMSE vs oracle {rmse:.2e} m, {flips} surfels beyond 1e-5 m, "
          f"bit-identical rows: {np.array_equal(got.view(np.uint32), ref.view(np.uint32))}")
    assert rmse <= 1e-5, rmse
    assert flips <= 1e-3 * n, flips
    # north-star gates above; with the pose sums defined (fixed point) and every lifecycle stage bit-exact, the two runs
    # are in fact the same bits
    assert np.array_equal(got_poses.astype(np.float32), ref_poses.astype(np.float32))
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


# ---- (e): BASELINE configs[4]'s resolution, joint BA with the intrinsics -------------------------------------------------
def test_c5_resolution_slice_with_intrinsics_matches_oracle():
    """A 20-keyframe slice of configs[4]: 1280 x 960 images (640 x 480 sparse cells: 300 blocks of append buffers in the
    intrinsics sweep, descriptor Jacobians 4 x those of 640 x 480 in the pose sums), every surfel the keyframes create, one
    iteration of the alternating scheme over geometry, poses, depth intrinsics + deformation and colour intrinsics."""
    ba, data, poses_gt, args = _bench_scene(width=1280, height=960, keyframes=20, surfels=10 ** 9)
    K = ba.keyframe_count()
    n = data.shape[1]
    assert n > 1000000, n
    ba.upload_surfels(data)
    # perturbed cameras, so that the intrinsics step has something to do (as tests/test_gpu_intrinsics_pcg_vs_oracle.py)
    cam = synthetic.test_camera(args.width, args.height).astype(np.float64)
    depth_cam = cam + np.array([0.5, -0.6, 1.23, -2.17])
    color_cam = cam + np.array([0.4, -0.3, 0.8, -0.6])
    ba.set_cameras(color_cam, depth_cam, 0.0)
    orc = _oracle_from_directba(ba, args, data)
    for name, values in (("depth_cam", depth_cam), ("color_cam", color_cam)):
        c = getattr(orc, name)
        c.fx, c.fy, c.cx, c.cy = [float(np.float32(v)) for v in values]
    ba.set_ba_iteration_counts(1, 1)               # equal counters: no end-of-scheme tasks (fixed surfel set)
    done, _ = ba.BundleAdjustment(optimize_depth_intrinsics=True, optimize_color_intrinsics=True, do_surfel_updates=False,
                                  optimize_poses=True, optimize_geometry=True, min_iterations=1, max_iterations=1,
                                  active_keyframe_window_start=0, active_keyframe_window_end=K - 1, increase_ba_iteration_count=False)
    orc.ba_iteration_count, orc.last_ba_iteration_count = 1, 1
    orc.use_depth, orc.use_desc = 1, 1
    stats = orc.bundle_adjustment(optimize_depth_intrinsics=True, optimize_color_intrinsics=True, do_surfel_updates=False,
                                  optimize_poses=True, optimize_geometry=True, min_iterations=1, max_iterations=1,
                                  increase_ba_iteration_count=False)
    assert done == stats.iterations_done == 1
    got = ba.download_surfels(8)
    ref = orc.surfel_data[:8, :n]
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    got_poses = np.array([ba.keyframe_pose(k) for k in range(K)])
    ref_poses = np.array([orc.pose(k) for k in range(K)])
    rmse = float(np.sqrt(np.mean(np.sum((got_poses[:, 4:] - ref_poses[:, 4:]) ** 2, axis=1))))
    assert rmse <= 1e-5, rmse                                                    # the north-star gate
    assert np.array_equal(got_poses.astype(np.float32), ref_poses.astype(np.float32))
    cc, dc, a = ba.cameras()
    ref_dc = np.array([orc.depth_cam.fx, orc.depth_cam.fy, orc.depth_cam.cx, orc.depth_cam.cy], np.float32)
    ref_cc = np.array([orc.color_cam.fx, orc.color_cam.fy, orc.color_cam.cx, orc.color_cam.cy], np.float32)
    moved = np.abs(ref_dc - depth_cam.astype(np.float32)).max()
    print(f"1280 x 960 slice: {n} surfels, pose RMSE vs oracle {rmse:.2e} m, depth camera moved by up to {moved:.3f} px, a = {orc.dp.a:.3e}")
    assert moved > 0.1                                                           # the step did something
    assert np.array_equal(np.asarray(dc, np.float32).view(np.uint32), ref_dc.view(np.uint32)), (dc, ref_dc)
    assert np.array_equal(np.asarray(cc, np.float32).view(np.uint32), ref_cc.view(np.uint32)), (cc, ref_cc)
    assert np.float32(a).view(np.uint32) == np.float32(orc.dp.a).view(np.uint32)
    assert np.array_equal(ba.cfactor().view(np.uint32), np.asarray(orc.cfactor, np.float32).view(np.uint32))


def test_c3_eight_surfel_shards_are_the_unsharded_run(c3):
    """BASELINE configs[2] sharded by surfels over EIGHT ranks -- what `bench.py --gpus 8` runs, on one GPU: eight vis::DirectBA
    instances (host threads) hold the 200 keyframes and every eighth chunk of 4096 of the 3 M surfels each and run
    BundleAdjustment(3 iterations) in lockstep; the all-reduce hook of the C ABI is served by an in-process loopback that sums the
    ranks' device buffers in a fixed order (RCCL refuses eight ranks on one device; the exchange is an integer sum either way).  Every
    rank must end with the poses of the unsharded call, bit for bit, and the union of the shards with its surfels.  At an eighth of the
    cloud (5 860 tiles) the ranks' geometry steps take the HYBRID launch shape and the pose sweeps share every tile among four
    wavefronts, neither of which the one-GPU run does: the sums' definitions (classes, fixed tree, fixed point) carry the identity."""
    import threading
    import torch
    from badslam_amd import multigpu
    from badslam_amd.directba import DirectBA
    ba, data, poses_gt, args = c3
    frames = args.frames
    assert frames is not None and len(frames) == args.keyframes
    K, N, WORLD, ITERATIONS = args.keyframes, data.shape[1], 8, 3
    lib = capi.load()
    start_poses = [ba.keyframe_pose(k) for k in range(K)]
    call = dict(do_surfel_updates=False, optimize_poses=True, optimize_geometry=True, min_iterations=ITERATIONS, max_iterations=ITERATIONS,
                increase_ba_iteration_count=False)
    # the unsharded call (device-driven loop)
    ba.upload_surfels(data)
    ba.set_ba_iteration_counts(1, 1)       # no end tasks: the surfel set and its order stay
    assert ba.BundleAdjustment(**call)[0] == ITERATIONS
    ref_poses = np.asarray([ba.keyframe_pose(k) for k in range(K)], np.float32)
    ref_surfels = ba.download_surfels(8)
    for k, T in enumerate(start_poses):     # leave the shared fixture as it was
        ba.set_keyframe_pose(k, T)
    ba.upload_surfels(data)

    cam = synthetic.test_camera(args.width, args.height)
    shards = [multigpu.shard_chunks(N, r, WORLD, 4096) for r in range(WORLD)]
    ranks = []
    for r in range(WORLD):
        rb = DirectBA(int(shards[r].size) + 4096, 1.0 / 5000, 40.0, args.cell, args.width, args.height, cam, cam)
        for (raw, rgb), T in zip(frames, poses_gt):
            rb.AddKeyframe(raw, rgb, T)
        for k, T in enumerate(start_poses):
            rb.set_keyframe_pose(k, T)
        rb.upload_surfels(np.ascontiguousarray(data[:, shards[r]]))
        rb.set_ba_iteration_counts(1, 1)
        ranks.append(rb)

    barrier, slots, errors, calls = threading.Barrier(WORLD), [None] * WORLD, [], [0]

    def hook_for(rank):
        def _hook(device_ptr, count, dtype, _stream, _user):
            try:
                torch.cuda.synchronize()
                slots[rank] = (device_ptr, count, dtype)
                barrier.wait(timeout=120)
                if rank == 0:
                    views = [torch.as_tensor(multigpu._DevicePtrView(p, n, d), device="cuda") for p, n, d in slots]
                    total = views[0].clone()
                    for v in views[1:]:
                        total += v
                    for v in views:
                        v.copy_(total)
                    torch.cuda.synchronize()
                    calls[0] += 1
                barrier.wait(timeout=120)
                return 0
            except Exception as e:   # noqa: BLE001 -- surfaces as a bahip error in the calling thread
                print("loopback all-reduce failed:", e, flush=True)
                barrier.abort()
                return 1
        return capi.ALLREDUCE_FN(_hook)

    hooks = [hook_for(r) for r in range(WORLD)]
    results = [None] * WORLD

    def rank_main(rank):
        try:
            torch.cuda.set_device(0)
            rb = ranks[rank]
            capi.check(lib.bahip_context_set_allreduce(rb.backend_context().handle, hooks[rank], None))
            done, _ = rb.BundleAdjustment(**call)
            results[rank] = dict(done=done, poses=np.asarray([rb.keyframe_pose(k) for k in range(K)], np.float32), surfels=rb.download_surfels(8))
        except Exception as e:   # noqa: BLE001
            errors.append((rank, repr(e)))
            barrier.abort()

    hybrid_before = C.c_longlong()
    capi.check(lib.bahip_debug_geometry_hybrid_launches(C.byref(hybrid_before)))
    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(WORLD)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors
    assert all(r is not None and r["done"] == ITERATIONS for r in results)
    hybrid_after = C.c_longlong()
    capi.check(lib.bahip_debug_geometry_hybrid_launches(C.byref(hybrid_after)))
    assert hybrid_after.value - hybrid_before.value >= WORLD * (ITERATIONS - 1)     # every rank, from its second iteration on
    assert calls[0] >= ITERATIONS                                                    # at least one exchange per pose phase
    merged = np.zeros_like(ref_surfels)
    for r in range(WORLD):
        assert np.array_equal(results[r]["poses"].view(np.uint32), ref_poses.view(np.uint32)), r
        merged[:, shards[r]] = results[r]["surfels"]
    assert np.array_equal(merged.view(np.uint32), ref_surfels.view(np.uint32))
    for rb in ranks:
        rb.close()
