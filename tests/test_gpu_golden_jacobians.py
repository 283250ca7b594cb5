"""The HIP kernels' residual Jacobian functions (ba_device.h: jac_*, the ones the sweeps call) on the golden vectors
generated from the reference's derivation script (tests/golden/jacobians.json), through bahip_debug_jacobian; and bit for bit
against the oracle's functions on the same inputs."""
import ctypes as C

import numpy as np
import pytest

from tests import golden_cases
from tests.test_cpu_golden_jacobians import _oracle, lib  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def test_hip_jacobians_match_golden_vectors_and_oracle_bits(lib):
    from badslam_amd import capi, lowlevel
    ctx = lowlevel.Context()
    cases = golden_cases.jacobian_cases()
    assert len(cases) >= 36
    for name, kind, x, expected, pick in cases:
        xin = np.asarray(x, np.float32)
        out = np.zeros(8, np.float32)
        capi.check(ctx.lib.bahip_debug_jacobian(ctx.handle, kind, xin.ctypes.data_as(C.POINTER(C.c_float)), len(xin),
                                                out.ctypes.data_as(C.POINTER(C.c_float)), 8))
        golden_cases.close([out[i] for i in pick], expected)
        ref = np.asarray(_oracle(lib, kind, [float(v) for v in xin]), np.float32)
        assert np.array_equal(out[:len(ref)].view(np.uint32), ref.view(np.uint32)), (name, out[:len(ref)], ref)
