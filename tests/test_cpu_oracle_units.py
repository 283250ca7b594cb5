"""CPU unit tests of the oracle's building blocks against independent numpy restatements:
packing formats, fp16 conversion, the software bilinear sampler, SE3 maps, depth calibration,
and the analytic Jacobians (central finite differences of independently written residual
functions in binary64)."""
import numpy as np
import pytest

from badslam_amd import se3, synthetic
from oracle import binding as ob
from tests import common

L = ob.lib()


def test_fp16_conversion_matches_ieee():
    rng = np.random.Generator(np.random.PCG64(0))
    vals = np.concatenate([rng.uniform(0, 1e-3, 20000), rng.uniform(0, 70000, 5000), 2.0 ** rng.uniform(-30, 17, 20000),
                           [0.0, 65504.0, 65519.9, 65520.0, 1e-8, 5.96e-8, 2.98e-8, 2.9802322e-8]]).astype(np.float32)
    for v in vals:
        h = L.orc_float_to_half(float(v))
        assert h == int(np.float16(v).view(np.uint16)), v
    for h in rng.integers(0, 0x7c00, 20000):
        assert L.orc_half_to_float(int(h)) == float(np.uint16(h).view(np.float16))


def test_normal_packing_roundtrip():
    rng = np.random.Generator(np.random.PCG64(1))
    import ctypes as C
    out = (C.c_float * 3)()
    for _ in range(2000):
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        L.orc_unpack_normal10(L.orc_pack_normal10(*[float(v) for v in n]), out)
        assert np.abs(np.array(list(out)) - n).max() < 2.5e-3      # 10-bit quantisation, renormalised on read
        assert abs(np.linalg.norm(list(out)) - 1) < 1e-6
        if n[2] < 0:                                               # image-space normals look at the camera
            L.orc_unpack_normal8(L.orc_pack_normal8(float(n[0]), float(n[1])), out)
            assert np.abs(np.array(list(out))[:2] - n[:2]).max() < 4.1e-3
            assert out[2] <= 0


def test_raw_to_calibrated_depth():
    # d = 1 / (1/(s raw) + c exp(-a/(s raw)))  (B/util.cuh:62-69)
    for a, c, s, raw in [(0.0, 0.0, 1 / 5000, 12345), (0.03, 0.005, 1 / 1000, 2500), (-0.01, -0.002, 1 / 5000, 40000)]:
        inv = 1.0 / (s * raw)
        ref = 1.0 / (inv + c * np.exp(-a * inv))
        assert L.orc_raw_to_calibrated_depth(a, c, s, raw) == pytest.approx(ref, rel=2e-6)


def test_bilinear_sampler_matches_clamped_reference():
    rng = np.random.Generator(np.random.PCG64(2))
    w, h = 37, 23
    img = np.zeros((h, w, 4), np.uint8)
    img[..., 3] = rng.integers(0, 256, size=(h, w))
    lum = img[..., 3].astype(np.float64) / 255.0

    def ref(x, y):   # CUDA linear filtering, unnormalised coords, clamp addressing (exact weights)
        xb, yb = x - 0.5, y - 0.5
        i, j = int(np.floor(xb)), int(np.floor(yb))
        a, b = xb - i, yb - j
        cl = lambda v, m: min(max(v, 0), m - 1)
        t = lambda xx, yy: lum[cl(yy, h), cl(xx, w)]
        return (1 - a) * (1 - b) * t(i, j) + a * (1 - b) * t(i + 1, j) + (1 - a) * b * t(i, j + 1) + a * b * t(i + 1, j + 1)

    pts = np.concatenate([rng.uniform(-3, w + 3, (3000, 1)), rng.uniform(-3, h + 3, (3000, 1))], axis=1)
    pts = np.concatenate([pts, [[0.5, 0.5], [w - 0.5, h - 0.5], [10.5, 7.5], [0, 0], [w, h]]])
    for x, y in pts:
        got = L.orc_sample_luma(img.ctypes.data, w, h, float(np.float32(x)), float(np.float32(y)))
        assert got == pytest.approx(ref(float(np.float32(x)), float(np.float32(y))), abs=3e-6)
    # texel centres return the texel exactly
    assert L.orc_sample_luma(img.ctypes.data, w, h, 10.5, 7.5) == pytest.approx(lum[7, 10], abs=1e-7)


def test_se3_matches_matrix_exponential():
    rng = np.random.Generator(np.random.PCG64(3))
    from scipy.linalg import expm
    for _ in range(200):
        xi = rng.uniform(-1, 1, 6) * rng.choice([1e-6, 1e-3, 0.5, 1.5])   # |omega| stays below pi (log is principal)
        T = ob.se3_exp(xi)
        M = np.eye(4)
        M[:3] = ob.se3_matrix3x4(T).reshape(3, 4)
        W = np.array([[0, -xi[5], xi[4], xi[0]], [xi[5], 0, -xi[3], xi[1]], [-xi[4], xi[3], 0, xi[2]], [0, 0, 0, 0]])
        assert np.abs(M - expm(W)).max() < 5e-6
        assert np.abs(ob.se3_log(T) - xi).max() < 2e-5 * max(1, np.abs(xi).max())
        I = ob.se3_mul(T, ob.se3_inverse(T))
        assert np.abs(ob.se3_log(I)).max() < 1e-5
    # python helper used by the scene generator agrees with the oracle
    xi = np.array([0.1, -0.2, 0.3, 0.4, 0.5, -0.6])
    assert np.abs(se3.matrix(se3.exp(xi))[:3] - ob.se3_matrix3x4(ob.se3_exp(xi)).reshape(3, 4)).max() < 1e-6


def test_ldlt_solve_matches_numpy():
    import ctypes as C
    rng = np.random.Generator(np.random.PCG64(4))
    for n in (4, 5, 6):
        A = rng.normal(size=(n, 2 * n))
        H = A @ A.T
        b = rng.normal(size=n)
        x = (C.c_double * n)()
        L.orc_ldlt_solve(n, (C.c_double * (n * n))(*H.ravel()), (C.c_double * n)(*b), x)
        assert np.allclose(list(x), np.linalg.solve(H, b), rtol=1e-9, atol=1e-12)
    # rank-deficient: pseudo-inverse behaviour on the null direction (like Eigen's LDLT solve)
    H = np.diag([2.0, 0.0, 3.0, 0.0])
    x = (C.c_double * 4)()
    L.orc_ldlt_solve(4, (C.c_double * 16)(*H.ravel()), (C.c_double * 4)(2, 5, 3, 7), x)
    assert list(x) == [1.0, 0.0, 1.0, 0.0]


# ---- Jacobians -------------------------------------------------------------------------------------------------------
def _scene_and_pairs():
    scene = common.small_scene(num_keyframes=2, seed=5, width=160, height=120)
    ba = common.build_oracle(scene, 50000)
    rng = np.random.Generator(np.random.PCG64(7))
    # move the second keyframe a little so residuals are non-trivial
    ba.set_pose(1, synthetic.perturb_pose(rng, scene.poses_gt[1], 0.004, 0.002))
    pairs = []
    for i in range(0, ba.surfels_size, 7):
        ok, e = ba.evaluate_pair(1, i)
        if ok and e.color_valid and 3 < e.px < 156 and 3 < e.py < 116:
            pairs.append((i, e))
    assert len(pairs) > 200
    return scene, ba, pairs


def _surfel(ba, i):
    import ctypes as C
    p = ba.surfel_data[0:3, i].astype(np.float64)
    n = (C.c_float * 3)()
    L.orc_unpack_normal10(int(ba.surfel_data[3, i].view(np.uint32)), n)
    return p, np.array(list(n), np.float64)


def test_depth_residual_pose_and_surfel_jacobians_by_finite_differences():
    scene, ba, pairs = _scene_and_pairs()
    fx, fy, cx, cy = [float(v) for v in scene.camera]
    T = ba.pose(1)
    eps = 1e-6
    worst_pose, worst_surf = 0.0, 0.0
    for i, e in pairs[:150]:
        p, n = _surfel(ba, i)
        d = float(e.calibrated_depth)
        u = d * np.array([(e.px - (cx - 0.5)) / fx, (e.py - (cy - 0.5)) / fy, 1.0])
        inv_std = float(e.depth_inv_stddev)

        def residual(pose, pos):   # fixed pixel, fixed sigma: what the reference's Jacobian assumes
            Fm = se3.matrix(se3.inverse(pose))
            l = Fm[:3, :3] @ pos + Fm[:3, 3]
            nl = Fm[:3, :3] @ n
            return inv_std * nl @ (u - l)

        assert residual(T, p) == pytest.approx(float(e.depth_residual), rel=2e-3, abs=2e-3)
        J = np.zeros(6)
        for c in range(6):
            xi = np.zeros(6); xi[c] = eps
            J[c] = (residual(se3.mul(T, se3.exp(xi)), p) - residual(se3.mul(T, se3.exp(-xi)), p)) / (2 * eps)
        ref = np.array(list(e.depth_jac_pose), np.float64)
        worst_pose = max(worst_pose, np.abs(J - ref).max() / max(1.0, np.abs(ref).max()))
        Js = (residual(T, p + eps * n) - residual(T, p - eps * n)) / (2 * eps)
        worst_surf = max(worst_surf, abs(Js - float(e.depth_jac_surfel)) / abs(float(e.depth_jac_surfel)))
    assert worst_pose < 2e-3, worst_pose
    assert worst_surf < 2e-3, worst_surf


def test_descriptor_pose_jacobian_is_gradient_times_projection_jacobian():
    """B/kernel_opt_pose.cu:122-141: J = (gx, gy) . d(pixel)/d(xi) for the right-multiplied update."""
    scene, ba, pairs = _scene_and_pairs()
    fx, fy, cx, cy = [float(v) for v in scene.camera]
    T = ba.pose(1)
    eps = 1e-6
    worst = 0.0
    for i, e in pairs[:150]:
        p, _ = _surfel(ba, i)

        def proj(pose):
            Fm = se3.matrix(se3.inverse(pose))
            l = Fm[:3, :3] @ p + Fm[:3, 3]
            return np.array([fx * l[0] / l[2] + cx, fy * l[1] / l[2] + cy])

        dpi = np.zeros((2, 6))
        for c in range(6):
            xi = np.zeros(6); xi[c] = eps
            dpi[:, c] = (proj(se3.mul(T, se3.exp(xi))) - proj(se3.mul(T, se3.exp(-xi)))) / (2 * eps)
        for t in range(2):
            g = np.array([e.grad[2 * t], e.grad[2 * t + 1]], np.float64)
            ref = np.array(list(e.desc_jac_pose[t]), np.float64)
            worst = max(worst, np.abs(g @ dpi - ref).max() / max(1.0, np.abs(ref).max()))
    assert worst < 2e-3, worst


def test_descriptor_surfel_jacobian_is_gradient_times_projection_jacobian():
    """B/kernel_opt_geometry.cu:188-192: moving the surfel along its normal."""
    scene, ba, pairs = _scene_and_pairs()
    fx, fy, cx, cy = [float(v) for v in scene.camera]
    Fm = se3.matrix(se3.inverse(ba.pose(1)))
    eps = 1e-6
    worst = 0.0
    for i, e in pairs[:150]:
        p, n = _surfel(ba, i)

        def proj(pos):
            l = Fm[:3, :3] @ pos + Fm[:3, 3]
            return np.array([fx * l[0] / l[2] + cx, fy * l[1] / l[2] + cy])

        dpi = (proj(p + eps * n) - proj(p - eps * n)) / (2 * eps)
        for t in range(2):
            g = np.array([e.grad[2 * t], e.grad[2 * t + 1]], np.float64)
            ref = float(e.desc_jac_surfel[t])
            # the reference's sign convention: position update is p -= x0 * n with J = -(g . dpi/dt)... verify magnitude + sign
            worst = max(worst, abs(g @ dpi - ref) / max(1.0, abs(ref)))
    assert worst < 2e-3, worst


def test_bilateral_filter_and_depth_cutoff_properties():
    """B/cuda_depth_processing.cu:42-128 restated: cutoff, holes, constancy, edge preservation."""
    rng = np.random.default_rng(3)
    s = 1.0 / 5000
    H, W = 48, 64
    depth = np.full((H, W), 10000, np.uint16)          # 2 m everywhere
    depth[5, 7] = 0                                     # hole
    depth[20:, 40:] = 20000                             # 4 m: beyond the 3 m cutoff
    depth[30, 10] = 10050                               # 1 cm bump
    out = ob.bilateral_filter_and_depth_cutoff(depth, 1.5, 0.005, 2.0, int(3.0 / s), s)
    assert out[5, 7] == 65535 and (out[20:, 40:] == 65535).all()           # unknown-depth marker
    far = out[:15, 12:30]                                                   # away from the hole and the cutoff region
    assert np.abs(far.astype(int) - 10000).max() <= 1                       # a constant stays constant (float round trip)
    assert 10000 <= out[30, 10] < 10050                                     # the bump is smoothed towards its neighbours
    # a step much larger than sigma_value in inverse depth survives: 1 m next to 2 m
    step = np.full((H, W), 10000, np.uint16)
    step[:, :32] = 5000
    out = ob.bilateral_filter_and_depth_cutoff(step, 1.5, 0.005, 2.0, 15000, s)
    assert np.abs(out[:, 31].astype(int) - 5000).max() <= 1 and np.abs(out[:, 32].astype(int) - 10000).max() <= 1
    # noise is reduced
    noisy = (10000 + rng.integers(-20, 21, (H, W))).astype(np.uint16)
    out = ob.bilateral_filter_and_depth_cutoff(noisy, 1.5, 0.005, 2.0, 15000, s)
    assert out[4:-4, 4:-4].astype(float).std() < 0.5 * noisy[4:-4, 4:-4].astype(float).std()


def test_assign_colors_is_the_mean_of_the_bilinear_samples():
    """orc_assign_colors (B/kernel_assign_colors.cu:41-125): on images of constant colour the mean of the bilinear samples is
    that colour exactly; with keyframes of different constant colours it lies between them and depends on which keyframes see
    the surfel; surfels no keyframe sees keep their colour; the luma channel of the RGBA sampler is the luma sampler."""
    import ctypes as C
    from oracle import binding as ob
    scene = common.small_scene(num_keyframes=3, width=160, height=120, seed=9)
    ba = common.build_oracle(scene, 60000)
    n = ba.surfels_size
    colors = ba.surfel_data[ob.SURFEL_COLOR if hasattr(ob, "SURFEL_COLOR") else 5, :n].view(np.uint8).reshape(n, 4)

    # (1) RGBA sampler, channel 3 == luma sampler, bit for bit
    L = ob.lib()
    L.orc_sample_luma.restype = C.c_float
    rgba = ba.kf_arrays(0)["color"]
    h, w = rgba.shape[:2]
    rng = np.random.default_rng(0)
    for x, y in rng.uniform(-2, max(w, h) + 2, (200, 2)):
        out = (C.c_float * 4)()
        L.orc_sample_rgba(rgba.ctypes.data_as(C.POINTER(C.c_uint8)), w, h, C.c_float(x), C.c_float(y), out)
        luma = L.orc_sample_luma(rgba.ctypes.data_as(C.POINTER(C.c_uint8)), w, h, C.c_float(x), C.c_float(y))
        assert np.float32(out[3]) == np.float32(luma)
        assert all(0.0 <= out[c] <= 1.0 for c in range(4))

    # (2) the same constant colour in every keyframe
    for k in range(3):
        ba.kf_arrays(k)["color"][:] = (10, 200, 77, 131)
    colors[:] = (1, 2, 3, 4)
    ba.assign_colors()
    seen = ~np.all(colors == (1, 2, 3, 4), axis=1)
    assert seen.mean() > 0.9
    assert np.all(colors[seen] == (10, 200, 77, 131))
    before = colors.copy()
    ba.assign_colors()
    assert np.array_equal(colors, before)                       # idempotent

    # (3) a different constant per keyframe: the result is a mean of a subset of them
    consts = np.array([(0, 30, 60, 90), (100, 130, 160, 190), (250, 240, 230, 220)], np.float64)
    for k in range(3):
        ba.kf_arrays(k)["color"][:] = consts[k].astype(np.uint8)
    ba.assign_colors()
    subsets = [[0], [1], [2], [0, 1], [0, 2], [1, 2], [0, 1, 2]]
    means = np.array([np.floor(consts[sub].mean(0) + 0.5) for sub in subsets])
    dist = np.abs(colors[seen].astype(np.float64)[:, None, :] - means[None]).max(2).min(1)
    assert dist.max() <= 1                                      # float rounding of the mean may differ by one level
    assert len(np.unique(colors[seen], axis=0)) >= 3            # several visibility patterns occur


def test_defined_sin_cos_atan_are_correctly_rounded_almost_everywhere():
    """orc_sincos / orc_atan (binary64 range reduction + polynomial, rounded to binary32): the SE(3) exponential and logarithm
    of both the oracle and the kernels use them instead of libm / the device library, so that pose updates are bit-identical.
    They must be as good as libm: within 1 ulp of the correctly rounded value, and equal to it but for rare ties."""
    import ctypes as C
    from oracle import binding as ob
    ob.lib()
    L = C.CDLL(ob._LIB_PATH)                   # a handle of its own: prototypes set here stay here
    L.orc_atan.restype = C.c_float
    L.orc_atan.argtypes = [C.c_float]
    L.orc_sincos.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-3.2, 3.2, 12000), rng.uniform(-1e-3, 1e-3, 3000), rng.uniform(-40, 40, 3000),
                        np.array([0.0, 1e-30, np.pi, -np.pi, np.pi / 2, np.pi / 4])]).astype(np.float32)
    sn, cs = C.c_float(), C.c_float()
    got = np.empty((3, x.size), np.float32)
    for i, v in enumerate(x):
        L.orc_sincos(float(v), C.byref(sn), C.byref(cs))
        got[:, i] = sn.value, cs.value, L.orc_atan(float(v))
    x64 = x.astype(np.float64)
    for row, exact in zip(got, (np.sin(x64), np.cos(x64), np.arctan(x64))):
        rounded = exact.astype(np.float32)
        a, b = row.view(np.int32).astype(np.int64), rounded.view(np.int32).astype(np.int64)
        a, b = np.where(a < 0, -(a & 0x7fffffff), a), np.where(b < 0, -(b & 0x7fffffff), b)
        ulps = np.abs(a - b)
        assert ulps.max() <= 1 and np.count_nonzero(ulps) <= 1e-3 * x.size, (int(ulps.max()), int(np.count_nonzero(ulps)))
