"""The oracle's residual Jacobians against golden vectors generated from the reference's own derivation script
(applications/badslam/scripts/jacobians_derivation.py, imported by scripts/make_golden_jacobians.py in the build container;
the vectors are committed in tests/golden/jacobians.json because /root/reference does not exist on the GPU box).

The golden side works in the reference's variables (global normal, global_T_frame, ...); the oracle's Jacobian functions take
what the kernels have at hand (normal and points in the keyframe frame, image gradients).  The conversion below is plain
linear algebra in binary64; the oracle evaluates in binary32, hence the 2e-5 relative tolerance."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from oracle import binding as ob

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jacobians.json")
F3, F4, F6 = C.c_float * 3, C.c_float * 4, C.c_float * 6


@pytest.fixture(scope="module")
def golden():
    with open(GOLDEN) as f:
        return json.load(f)["cases"]


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


def _close(got, want, rel=2e-5):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert np.abs(got - want).max() <= rel * max(1.0, np.abs(want).max()), (got, want)


def _bilinear_gradient(texels, px, py):
    """Derivative of the bilinear lookup wrt the pixel position (what B/cost_function.cuh:200-211 samples)."""
    tl, tr, bl, br = texels
    fx, fy = px - np.floor(px), py - np.floor(py)
    return (1 - fy) * (tr - tl) + fy * (br - bl), (1 - fx) * (bl - tl) + fx * (br - tr)


def test_golden_file_comes_from_the_reference_script(golden):
    meta = json.load(open(GOLDEN))
    assert "jacobians_derivation.py" in meta["source"] and meta["generator"] == "scripts/make_golden_jacobians.py"
    assert all(len(v) >= 6 for v in golden.values())


def test_depth_residual_pose_jacobian(golden, lib):
    for c in golden["depth_pose"]:
        G = np.array(c["global_T_frame"]).reshape(3, 4)
        nl = G[:, :3].T @ np.array(c["surfel_normal"])           # surfel normal in the keyframe frame
        J = F6()
        lib.orc_jac_depth_pose(F3(*nl), F3(*c["local_point"]), 1.0, J)
        _close(list(J), c["jacobian"])


def test_depth_residual_surfel_jacobian(golden):
    # d/dt of n . (g - (s + t n)) = -|n|^2 = -1: the kernels use -inv_sigma (B/kernel_opt_geometry.cu:143-146)
    for c in golden["depth_surfel"]:
        _close(c["jacobian"], [-1.0], rel=1e-12)


def test_depth_residual_intrinsics_and_correction_jacobians(golden, lib):
    for c in golden["depth_intrinsics"]:
        G = np.array(c["global_T_frame"]).reshape(3, 4)
        nl = G[:, :3].T @ np.array(c["surfel_normal"])
        J = F6()
        lib.orc_jac_depth_intrinsics(c["x"], c["y"], c["depth"], 1.0, nl[0], nl[1], 0.0, 0.0, 1.0, 1.0, 1.0, J)
        _close(list(J)[:4], c["jacobian"])                       # fx_inv, fy_inv, cx_inv, cy_inv
    for c in golden["depth_correction"]:
        G = np.array(c["global_T_frame"]).reshape(3, 4)
        nl = G[:, :3].T @ np.array(c["surfel_normal"])
        fx_inv, fy_inv, cx_inv, cy_inv = c["intrinsics"]
        nx, ny = fx_inv * c["x"] + cx_inv, fy_inv * c["y"] + cy_inv
        exp_inv_depth = np.exp(-c["a"] * c["raw_inv_depth"])
        corrected = c["cfactor"] * exp_inv_depth + c["raw_inv_depth"]
        J = F6()
        lib.orc_jac_depth_intrinsics(c["x"], c["y"], 1.0 / corrected, 1.0, nl[0], nl[1], float(np.dot([nx, ny, 1.0], nl)), c["cfactor"],
                                     c["raw_inv_depth"], exp_inv_depth, corrected, J)
        _close([J[5], J[4]], c["jacobian"])                      # golden order: cfactor, a; oracle rows: [4] = a, [5] = cfactor


def test_descriptor_residual_pose_jacobian(golden, lib):
    for c in golden["descriptor_pose"]:
        ls = np.array(c["local_surfel_pos"])
        fx, fy, cx, cy = c["camera"]
        gx, gy = _bilinear_gradient(c["texels"], fx * ls[0] / ls[2] + cx, fy * ls[1] / ls[2] + cy)
        J = F6()
        lib.orc_jac_descriptor_pose(F3(*ls), gx * fx, gy * fy, J)   # the kernels pass the gradient times fx, fy
        _close(list(J), c["jacobian"])


def test_descriptor_residual_surfel_jacobian(golden, lib):
    for c in golden["descriptor_surfel"]:
        Fm = np.array(c["frame_T_global"]).reshape(3, 4)
        lp = Fm[:, :3] @ np.array(c["surfel_pos"]) + Fm[:, 3]
        rn = Fm[:, :3] @ np.array(c["surfel_normal"])
        fx, fy, cx, cy = c["camera"]
        gx, gy = _bilinear_gradient(c["texels"], fx * lp[0] / lp[2] + cx, fy * lp[1] / lp[2] + cy)
        got = lib.orc_jac_descriptor_surfel(F3(*rn), F3(*lp), gx, gy, fx, fy)
        _close([got], c["jacobian"])


def test_descriptor_residual_color_intrinsics_jacobian(golden, lib):
    for c in golden["descriptor_color_intrinsics"]:
        ls = np.array(c["local_surfel_pos"])
        fx, fy, cx, cy = c["camera"]
        gx, gy = _bilinear_gradient(c["texels"], fx * ls[0] / ls[2] + cx, fy * ls[1] / ls[2] + cy)
        J = F4()
        lib.orc_jac_descriptor_color_intrinsics(gx, gy, ls[0] / ls[2], ls[1] / ls[2], J)
        _close(list(J), c["jacobian"])
