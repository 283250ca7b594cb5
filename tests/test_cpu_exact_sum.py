"""The exact accumulator of the oracle (oracle_exact.c) against Python's math.fsum, which returns the correctly rounded
binary64 sum of its inputs by construction: an independent pin of the definition the PCG scheme's dense sums and dot products
rest on (the backend's device-side accumulator is held against the same reference in tests/test_gpu_exact_sum.py)."""
import math

import numpy as np
import pytest

from oracle import binding as ob


def _cases():
    rng = np.random.Generator(np.random.PCG64(5))
    yield "empty", np.zeros(0, np.float32)
    yield "zeros", np.array([0.0, -0.0, 0.0], np.float32)
    yield "one", np.array([1.5], np.float32)
    yield "ones", np.ones(100000, np.float32)
    yield "normal", rng.standard_normal(200000).astype(np.float32)
    # 60 binades of dynamic range, both signs: a binary64 running sum is order dependent here
    wide = (rng.standard_normal(100000) * np.exp2(rng.integers(-30, 30, 100000))).astype(np.float32)
    yield "wide", wide
    # massive cancellation: the sum is 2^-140 scale while the terms reach 2^120
    big = (rng.standard_normal(5000) * np.exp2(rng.integers(60, 120, 5000))).astype(np.float32)
    yield "cancel", np.concatenate([big, -big, np.array([1e-42, 3e-45], np.float32)])
    # denormals and the extremes of the exponent range
    tiny = np.frombuffer(rng.integers(1, 1 << 23, 4096, dtype=np.uint32).tobytes(), np.float32)
    yield "denormal", np.concatenate([tiny, -tiny[:1000]])
    yield "extremes", np.array([np.finfo(np.float32).max] * 3000 + [np.float32(1.4e-45)] * 7 + [-np.finfo(np.float32).max] * 1000, np.float32)
    # ties: sums that fall exactly between two binary64 values must round to even
    yield "tie_even", np.array([2.0 ** 60, 2.0 ** 7, 2.0 ** 7 * 0], np.float32)          # 2^60 + 2^7 is exactly half an ulp(2^60) = 2^8 -> tie
    yield "tie_odd", np.array([2.0 ** 60, 2.0 ** 8, 2.0 ** 7], np.float32)               # (2^60 + 2^8) + half an ulp: odd -> rounds up
    yield "tie_sticky", np.array([2.0 ** 60, 2.0 ** 7, 2.0 ** -100], np.float32)         # just above the tie -> rounds up
    yield "negative_tie", np.array([-(2.0 ** 60), -(2.0 ** 7)], np.float32)
    for n in (1, 2, 3, 63, 64, 65, 1000):
        yield f"random_bits_{n}", np.frombuffer(rng.integers(0, 1 << 32, n, dtype=np.uint32).tobytes(), np.float32)


@pytest.mark.parametrize("name,values", [(n, v) for n, v in _cases()], ids=[n for n, _ in _cases()])
def test_exact_sum_is_the_correctly_rounded_sum(name, values):
    values = np.asarray(values, np.float32)
    got = ob.exact_sum(values)
    if not np.all(np.isfinite(values)):
        assert math.isnan(got)
        return
    want = math.fsum(float(v) for v in values)
    assert got == want and math.copysign(1.0, got) == math.copysign(1.0, want if want != 0 else 1.0), (name, got, want)


def test_exact_sum_does_not_depend_on_the_order():
    rng = np.random.Generator(np.random.PCG64(6))
    v = (rng.standard_normal(50000) * np.exp2(rng.integers(-40, 40, 50000))).astype(np.float32)
    ref = ob.exact_sum(v)
    for _ in range(3):
        assert ob.exact_sum(rng.permutation(v)) == ref
    # while a plain binary32 or binary64 running sum does
    assert len({float(np.cumsum(rng.permutation(v), dtype=np.float64)[-1]) for _ in range(6)}) > 1


def test_non_finite_terms_make_the_sum_nan():
    assert math.isnan(ob.exact_sum(np.array([1.0, np.inf, -np.inf], np.float32)))
    assert math.isnan(ob.exact_sum(np.array([1.0, np.nan], np.float32)))
