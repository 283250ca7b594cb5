"""The oracle's residual Jacobians against golden vectors generated from the reference's own derivation script
(applications/badslam/scripts/jacobians_derivation.py, imported by scripts/make_golden_jacobians.py in the build container;
the vectors are committed in tests/golden/jacobians.json because /root/reference does not exist on the GPU box).
tests/golden_cases.py converts the reference's variables into the arguments of the Jacobian functions; the oracle evaluates
in binary32, hence the 2e-5 relative tolerance.  tests/test_gpu_golden_jacobians.py runs the same cases through the HIP
functions."""
import ctypes as C

import numpy as np
import pytest

from oracle import binding as ob
from tests import golden_cases

F3, F4, F6 = C.c_float * 3, C.c_float * 4, C.c_float * 6


@pytest.fixture(scope="module")
def lib():
    L = ob.lib()
    L.orc_jac_depth_pose.argtypes = [F3, F3, C.c_float, F6]
    L.orc_jac_descriptor_pose.argtypes = [F3, C.c_float, C.c_float, F6]
    L.orc_jac_descriptor_surfel.argtypes = [F3, F3, C.c_float, C.c_float, C.c_float, C.c_float]
    L.orc_jac_descriptor_surfel.restype = C.c_float
    L.orc_jac_depth_intrinsics.argtypes = [C.c_int, C.c_int] + [C.c_float] * 9 + [F6]
    L.orc_jac_descriptor_color_intrinsics.argtypes = [C.c_float] * 4 + [F4]
    return L


def _oracle(L, kind, x):
    if kind == 0:
        J = F6(); L.orc_jac_depth_pose(F3(*x[0:3]), F3(*x[3:6]), x[6], J); return list(J)
    if kind == 1:
        J = F6(); L.orc_jac_descriptor_pose(F3(*x[0:3]), x[3], x[4], J); return list(J)
    if kind == 2:
        return [L.orc_jac_descriptor_surfel(F3(*x[0:3]), F3(*x[3:6]), x[6], x[7], x[8], x[9])]
    if kind == 3:
        J = F6(); L.orc_jac_depth_intrinsics(int(x[0]), int(x[1]), *x[2:11], J); return list(J)
    J = F4(); L.orc_jac_descriptor_color_intrinsics(*x[0:4], J); return list(J)


def test_golden_file_comes_from_the_reference_script():
    meta = golden_cases.load()
    assert "jacobians_derivation.py" in meta["source"] and meta["generator"] == "scripts/make_golden_jacobians.py"
    assert all(len(v) >= 6 for v in meta["cases"].values())


def test_depth_residual_surfel_jacobian_is_minus_one():
    # d/dt of n . (g - (s + t n)) = -|n|^2 = -1: the kernels use -inv_sigma (B/kernel_opt_geometry.cu:143-146)
    for c in golden_cases.load()["cases"]["depth_surfel"]:
        golden_cases.close(c["jacobian"], [-1.0], rel=1e-12)


@pytest.mark.parametrize("name", ["depth_pose", "depth_intrinsics", "depth_correction", "descriptor_pose", "descriptor_surfel",
                                  "descriptor_color_intrinsics"])
def test_oracle_jacobian_matches_the_reference_derivation(lib, name):
    cases = [c for c in golden_cases.jacobian_cases() if c[0] == name]
    assert len(cases) >= 6
    for _, kind, x, expected, pick in cases:
        got = _oracle(lib, kind, [float(v) for v in x])
        golden_cases.close([got[i] for i in pick], expected)
