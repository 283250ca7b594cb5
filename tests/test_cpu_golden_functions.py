"""The oracle's building blocks against golden VALUES generated from the reference's own derivation script
(applications/badslam/scripts/jacobians_derivation.py, imported by scripts/make_golden_functions.py in the build container;
the vectors are committed in tests/golden/functions.json because /root/reference does not exist on the GPU box): depth
calibration, pinhole projection, pixel-centre unprojection, the weights of the bilinear sampler, and the rotation of the
exponential map.  The oracle evaluates in binary32 (2e-6 relative); the kernels are bit-identical to the oracle on these
functions (tests/test_gpu_scale_parity.py::test_c3_sampled_pairs_bit_exact, test_pose_update_step_bit_exact)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from oracle import binding as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "functions.json")) as f:
        doc = json.load(f)
    assert "jacobians_derivation.py" in doc["source"] and doc["generator"] == "scripts/make_golden_functions.py"
    assert all(len(v) >= 16 for v in doc["cases"].values())
    return doc["cases"]


@pytest.fixture(scope="module")
def L():
    """A handle of its own: the prototypes set below must not leak into the cached handle the other tests share."""
    ob.lib()                                   # builds the library if it is not there yet
    return C.CDLL(ob._LIB_PATH)


def _close(got, expected, rel=2e-6, floor=0.0):
    got, expected = np.asarray(got, np.float64), np.asarray(expected, np.float64)
    assert np.all(np.abs(got - expected) <= rel * np.maximum(np.abs(expected), floor)), (got, expected)


def test_depth_calibration(golden, L):
    """CorrectDepth(cfactor, a, 1 / (s raw)) = orc_raw_to_calibrated_depth (B/util.cuh:62-69)."""
    L.orc_raw_to_calibrated_depth.restype = C.c_float
    L.orc_raw_to_calibrated_depth.argtypes = [C.c_float, C.c_float, C.c_float, C.c_uint16]
    for c in golden["correct_depth"]:
        _close(L.orc_raw_to_calibrated_depth(c["a"], c["cfactor"], c["raw_to_float_depth"], c["raw"]), c["value"])


class _V3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


def test_projection(golden, L):
    """Project(point, fx, fy, cx, cy): the projection every sweep uses, reached through the tangent-point projection with a
    zero radius (both tangent points coincide with the surfel) and an identity pose."""
    L.orc_tangent_projections.restype = None
    L.orc_tangent_projections.argtypes = [_V3, _V3, C.c_float, C.POINTER(C.c_float), C.POINTER(ob.Camera), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    identity = (C.c_float * 12)(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0)
    for c in golden["project"]:
        cam = ob.make_camera(c["camera"], 640, 480)
        t1, t2 = (C.c_float * 2)(), (C.c_float * 2)()
        L.orc_tangent_projections(_V3(*c["point"]), _V3(0.0, 0.0, -1.0), 0.0, identity, C.byref(cam), t1, t2)
        _close(list(t1), c["value"])
        _close(list(t2), c["value"])


def test_unprojection(golden, L):
    """Unproject(x, y, depth, 1/fx, 1/fy, -(cx - 0.5)/fx, -(cy - 0.5)/fy): PixelCenterUnprojector (B/surfel_projection.cuh:88-126)."""
    L.orc_unproject.restype = None
    L.orc_unproject.argtypes = [C.POINTER(ob.Camera), C.c_int, C.c_int, C.c_float, C.POINTER(C.c_float)]
    for c in golden["unproject"]:
        cam = ob.make_camera(c["camera"], 640, 480)
        out = (C.c_float * 3)()
        L.orc_unproject(C.byref(cam), c["x"], c["y"], c["depth"], out)
        # near the principal point fx_inv * x + cx_inv cancels: the error scales with the depth, not with the coordinate
        _close(list(out), c["value"], rel=4e-7, floor=c["depth"])


def test_bilinear_weights(golden, L):
    """InterpolateBilinear(fx, fy, TL, TR, BL, BR) = the sampler at (0.5 + fx, 0.5 + fy) of a 2 x 2 image (texel centres at
    integer + 0.5, B/keyframe.cc:67-73), luma in the fourth channel, normalised by 255."""
    L.orc_sample_luma.restype = C.c_float
    L.orc_sample_luma.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_float, C.c_float]
    for c in golden["bilinear"]:
        rgba = np.zeros((2, 2, 4), np.uint8)
        rgba[0, 0, 3], rgba[0, 1, 3], rgba[1, 0, 3], rgba[1, 1, 3] = c["texels"]
        got = L.orc_sample_luma(rgba.ctypes.data_as(C.POINTER(C.c_uint8)), 2, 2, 0.5 + c["x"], 0.5 + c["y"])
        _close(got, c["value"], floor=1e-2)


def test_rotation_of_the_exponential_map(golden, L):
    """SO3exp(omega) -> QuaternionToRotationMatrix: the rotation block of orc_se3_exp for small rotations.  The script
    implements a small-angle branch that is only meant to be differentiated at zero (its real part reads 1 - theta^2 / 2
    where the half-angle cosine is 1 - theta^2 / 8), so the vectors use |omega| <= 3.5e-3, where it is exact to 1e-8."""
    L.orc_se3_exp.restype = None
    L.orc_se3_matrix3x4.restype = None
    for c in golden["so3_exp"]:
        tangent = (C.c_float * 6)(0.0, 0.0, 0.0, *c["omega"])
        T = ob.SE3()
        L.orc_se3_exp(tangent, C.byref(T))
        m = (C.c_float * 12)()
        L.orc_se3_matrix3x4(C.byref(T), m)
        R = [m[4 * i + j] for i in range(3) for j in range(3)]
        assert np.abs(np.asarray(R, np.float64) - np.asarray(c["value"])).max() < 1.5e-7, (R, c["value"])
        assert [m[3], m[7], m[11]] == [0.0, 0.0, 0.0]
