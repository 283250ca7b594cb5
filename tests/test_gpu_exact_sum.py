"""The device-side exact accumulator (badslam_amd/csrc/exact_sum.h) against Python's math.fsum -- the correctly rounded
binary64 sum by construction -- and against the oracle's restatement: the definition of the PCG scheme's dense sums and
dot products, pinned independently of both implementations."""
import math

import numpy as np
import pytest

from oracle import binding as ob
from tests import common
from tests.test_cpu_exact_sum import _cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    scene = common.small_scene(num_keyframes=1, seed=3)
    return common.build_gpu(scene, 1000, create_from=[])


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("name,values", [(n, v) for n, v in _cases()], ids=[n for n, _ in _cases()])
def test_device_exact_sum_is_the_correctly_rounded_sum(gpu, name, values, mode):
    values = np.asarray(values, np.float32)
    got = gpu.exact_sum(values, mode)
    if not np.all(np.isfinite(values)):
        assert math.isnan(got)
        return
    want = math.fsum(float(v) for v in values)
    assert got == want, (name, mode, got, want)
    assert got == ob.exact_sum(values)


@pytest.mark.parametrize("mode", [0, 1])
def test_device_exact_sum_large_and_order_free(gpu, mode):
    rng = np.random.Generator(np.random.PCG64(8))
    v = (rng.standard_normal(3_000_000) * np.exp2(rng.integers(-25, 25, 3_000_000))).astype(np.float32)
    want = math.fsum(v.astype(np.float64))
    assert gpu.exact_sum(v, mode) == want
    assert gpu.exact_sum(v[::-1].copy(), mode) == want


def test_non_finite_terms_resolve_to_nan(gpu):
    for mode in (0, 1):
        assert math.isnan(gpu.exact_sum(np.array([1.0, np.inf], np.float32), mode))
        assert math.isnan(gpu.exact_sum(np.array([np.nan, 2.0], np.float32), mode))
        assert gpu.exact_sum(np.array([1.0, 2.0], np.float32), mode) == 3.0   # the flag does not stick across calls
