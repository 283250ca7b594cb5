"""rcp_exact / sqrt_exact of ba_device.h (hardware approximation + one fused correction step, 5 and 8 instructions instead
of the compiler's 11 and 16) must equal IEEE `1.f / x` and `sqrtf(x)` -- what the oracle computes -- for EVERY binary32
significand: exhaustive over all 2^23 significands in several binades, plus the special values."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(ctx, kind, x):
    from badslam_amd import capi
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    capi.check(ctx.lib.bahip_debug_exact_math(ctx.handle, kind, x.ctypes.data_as(C.POINTER(C.c_float)),
                                              out.ctypes.data_as(C.POINTER(C.c_float)), x.size))
    return out


@pytest.fixture(scope="module")
def ctx():
    from badslam_amd import lowlevel
    return lowlevel.Context()


@pytest.mark.parametrize("exponent", [-20, -3, -1, 0, 1, 2, 13, 40, -60, 100])
def test_reciprocal_is_correctly_rounded_for_every_significand(ctx, exponent):
    bits = (np.arange(1 << 23, dtype=np.uint32) | np.uint32((127 + exponent) << 23))
    for sign in (0, 0x80000000):
        x = (bits | np.uint32(sign)).view(np.float32)
        got = _run(ctx, 0, x)
        ref = (np.float32(1.0) / x).astype(np.float32)
        bad = np.flatnonzero(got.view(np.uint32) != ref.view(np.uint32))
        assert bad.size == 0, (exponent, sign, bad.size, x[bad[:5]], got[bad[:5]], ref[bad[:5]])


def test_reciprocal_special_values(ctx):
    x = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, -1.0, 3.0, 1e-30, 1e30], np.float32)
    got = _run(ctx, 0, x)
    with np.errstate(divide="ignore"):
        ref = (np.float32(1.0) / x).astype(np.float32)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    ok = ~np.isnan(ref)
    assert np.array_equal(got[ok].view(np.uint32), ref[ok].view(np.uint32))


@pytest.mark.parametrize("exponent", [-60, -24, -23, -2, -1, 0, 1, 2, 3, 30])
def test_square_root_is_correctly_rounded_for_every_significand(ctx, exponent):
    x = (np.arange(1 << 23, dtype=np.uint32) | np.uint32((127 + exponent) << 23)).view(np.float32)
    got = _run(ctx, 1, x)
    ref = np.sqrt(x).astype(np.float32)
    bad = np.flatnonzero(got.view(np.uint32) != ref.view(np.uint32))
    assert bad.size == 0, (exponent, bad.size, x[bad[:5]], got[bad[:5]], ref[bad[:5]])


def test_square_root_of_every_packed_normal(ctx):
    """The only hot-loop use: z = -sqrt(max(0, 1 - x^2 - y^2)) of the 2 x s8 packed measurement normals (all 65536 codes)."""
    i, j = np.meshgrid(np.arange(-128, 128), np.arange(-128, 128), indexing="ij")
    x = (i.astype(np.float32) * np.float32(1.0 / 127.0)).astype(np.float32)
    y = (j.astype(np.float32) * np.float32(1.0 / 127.0)).astype(np.float32)
    z = (np.float32(1) - x * x - y * y).astype(np.float32)
    z = np.where(z > 0, z, np.float32(0)).astype(np.float32).ravel()
    got = _run(ctx, 1, z)
    assert np.array_equal(got.view(np.uint32), np.sqrt(z).astype(np.float32).view(np.uint32))
    assert got[z == 0].size > 0 and np.all(got[z == 0] == 0)


# ---- the defined sin / cos / atan of the SE(3) exponential and logarithm (se3_device.h: sincos_det, atan_det) ------------------
def _ulp_distance(a, b):
    ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    ia, ib = np.where(ia < 0, -(ia & 0x7fffffff), ia), np.where(ib < 0, -(ib & 0x7fffffff), ib)
    return np.abs(ia - ib)


def _trig_samples():
    rng = np.random.default_rng(17)
    return np.concatenate([rng.uniform(-3.2, 3.2, 400000), rng.uniform(-1e-3, 1e-3, 100000), rng.uniform(-40, 40, 100000),
                           np.array([0.0, -0.0, 1e-30, np.pi, -np.pi, np.pi / 2, np.pi / 4])]).astype(np.float32)


def test_defined_trigonometry_equals_the_oracle_and_is_correctly_rounded(ctx):
    """Pose updates are exp / log of SE(3); device library and glibc differ in the last bit of sin / cos / atan now and then,
    which would end bit parity of the poses.  Both sides therefore evaluate the same binary64 range reduction + polynomial,
    rounded to binary32: device == oracle on every sample, and both within 1 ulp of the correctly rounded value (binary64
    libm rounded to binary32), almost always equal to it."""
    import ctypes as C
    from oracle import binding as ob
    ob.lib()
    L = C.CDLL(ob._LIB_PATH)                   # a handle of its own: prototypes set here stay here
    L.orc_atan.restype = C.c_float
    L.orc_atan.argtypes = [C.c_float]
    L.orc_sincos.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    x = _trig_samples()
    sn, cs = C.c_float(), C.c_float()
    ref_sin, ref_cos, ref_atan = np.empty_like(x), np.empty_like(x), np.empty_like(x)
    for i, v in enumerate(x[::37]):      # the oracle is called value by value through ctypes: a 1/37 sample is plenty
        L.orc_sincos(float(v), C.byref(sn), C.byref(cs))
        ref_sin[i], ref_cos[i], ref_atan[i] = sn.value, cs.value, L.orc_atan(float(v))
    n = len(x[::37])
    for kind, ref, exact in ((2, ref_sin, np.sin), (3, ref_cos, np.cos), (4, ref_atan, np.arctan)):
        got = _run(ctx, kind, x)
        assert np.array_equal(got[::37].view(np.uint32), ref[:n].view(np.uint32)), kind
        rounded = exact(x.astype(np.float64)).astype(np.float32)
        ulps = _ulp_distance(got, rounded)
        assert ulps.max() <= 1, (kind, int(ulps.max()))
        assert np.count_nonzero(ulps) <= 1e-4 * x.size, (kind, int(np.count_nonzero(ulps)))


def test_defined_exponential_equals_the_oracle_and_is_correctly_rounded(ctx):
    """exp(-a / depth), the depth deformation (ba_device.h: exp_det / the oracle's orc_exp): the same binary32 reduction and
    polynomial on both sides -- device == oracle on every sample --, within 2 ulp of the correctly rounded value (what CUDA
    documents for the reference's expf); special values as expf has them."""
    import ctypes as C
    from oracle import binding as ob
    ob.lib()
    L = C.CDLL(ob._LIB_PATH)
    L.orc_exp.restype = C.c_float
    L.orc_exp.argtypes = [C.c_float]
    rng = np.random.default_rng(23)
    x = np.concatenate([rng.uniform(-2.0, 2.0, 400000), rng.uniform(-1e-3, 1e-3, 100000), rng.uniform(-100, 88, 100000),
                        np.array([0.0, -0.0, 1e-30, -1e-30, 88.7, -87.3, -103.9, 0.34657359, -0.34657359])]).astype(np.float32)
    got = _run(ctx, 5, x)
    ref = np.array([L.orc_exp(float(v)) for v in x[::37]], np.float32)
    assert np.array_equal(got[::37].view(np.uint32), ref.view(np.uint32))
    with np.errstate(over="ignore", under="ignore"):
        rounded = np.exp(x.astype(np.float64)).astype(np.float32)
    normal = rounded > np.float32(1.2e-38)             # (subnormal results: ldexp rounds twice; the deformation never gets there)
    ulps = _ulp_distance(got[normal], rounded[normal])
    assert ulps.max() <= 2, int(ulps.max())
    assert np.count_nonzero(ulps > 1) <= 1e-3 * x.size, int(np.count_nonzero(ulps > 1))
    special = np.array([np.nan, np.inf, -np.inf, 200.0, -200.0], np.float32)
    got = _run(ctx, 5, special)
    assert np.isnan(got[0]) and np.isposinf(got[1]) and got[2] == 0 and np.isposinf(got[3]) and got[4] == 0
