"""Closed-loop pins of the oracle against the reference's own pose tests.

Restates applications/badslam/src/badslam/test/test_pose_optimization_geometric_residual.cc:50-178
and test_pose_optimization_photometric_residual.cc:50-185 against the CPU oracle: same scene
structure, same 13 start offsets, same pass tolerances (1.1e-6 / 8e-5 on every component of
log(T_est^-1 * T_gt)).  Scene content uses a portable PRNG instead of glibc rand()/Eigen::Random.
"""
import numpy as np
import pytest

from badslam_amd import se3, synthetic
from oracle import binding as ob

W, H = 640, 480


def _offsets(kt, kr):
    out = [np.zeros(6)]
    for sign in (1, -1):
        for i in range(3):
            v = np.zeros(6); v[i] = sign * kt; out.append(v)
        for i in range(3):
            v = np.zeros(6); v[3 + i] = sign * kr; out.append(v)
    return out


def _three_planes_depth(rng, cam, s):
    fx, fy, cx, cy = [float(v) for v in cam]
    depth = np.full((H, W), 65535, dtype=np.uint16)
    for p in range(3):
        n = rng.uniform(-1, 1, 3); n[2] = -1.0; n /= np.linalg.norm(n)
        max_x, min_x = W - 10 - 1, 10
        left = int(min_x + (max_x - min_x) * ((2 * p) / (2.0 * 3 - 1)))
        right = int(min_x + (max_x - min_x) * ((2 * p + 1) / (2.0 * 3 - 1)))
        ys, xs = np.mgrid[10:H - 10, left:right]
        dirs = np.stack([(xs - (cx - 0.5)) / fx, (ys - (cy - 0.5)) / fy, np.ones_like(xs, dtype=np.float64)], -1)
        z = -2.5 / (dirs @ n)
        depth[10:H - 10, left:right] = (z / s + 0.5).astype(np.uint16)
    return depth


def test_pose_optimization_geometric_residual():
    rng = np.random.Generator(np.random.PCG64(0))
    cam = synthetic.test_camera(W, H)
    s = 1.0 / 1000
    c = ob.make_camera(cam, W, H)
    ba = ob.OracleBA(1000 * 1000, s, 40.0, 1, c, c, use_depth_residuals=True, use_descriptor_residuals=False)
    depth = _three_planes_depth(rng, cam, s)
    rgb = np.zeros((H, W, 3), np.uint8)
    gt = se3.identity()
    ba.add_keyframe(depth, rgb, gt)
    n = ba.create_surfels_for_keyframe(0, filter_new_surfels=False)
    assert n > 100000
    gt_c = ob.SE3.from_array(gt)
    worst = 0.0
    for xi in _offsets(0.005, 0.001):
        init = ob.se3_mul(ob.se3_exp(xi), ob.se3_inverse(gt_c))
        est, its, conv = ba.estimate_frame_pose(0, init)
        err = ob.se3_log(ob.se3_mul(ob.se3_inverse(est), gt_c))
        worst = max(worst, np.abs(err).max())
        assert conv
    assert worst <= 1.1e-6, worst


def _smooth_random_rgb(rng):
    img = np.zeros((H, W, 3), np.int32)
    noise = rng.integers(0, 16, size=(H, W))
    for y in range(H):
        top = img[y - 1] if y > 0 else np.zeros((W, 3), np.int32)
        row = img[y]
        for x in range(1, W):
            row[x] = ((row[x - 1] + top[x]) // 2 + noise[y, x]) & 0xff
    return img.astype(np.uint8)


def test_pose_optimization_color_only_cues():
    rng = np.random.Generator(np.random.PCG64(0))
    cam = synthetic.test_camera(W, H)
    s = 1.0 / 1000
    c = ob.make_camera(cam, W, H)
    ba = ob.OracleBA(1000 * 1000, s, 40.0, 1, c, c, use_depth_residuals=False, use_descriptor_residuals=True)
    depth = np.full((H, W), int(2 / s), dtype=np.uint16)
    depth[0, :] = 65535; depth[-1, :] = 65535; depth[:, 0] = 65535; depth[:, -1] = 65535
    rgb = _smooth_random_rgb(rng)
    gt_c = ob.se3_exp([0.1, 0.2, 0.3, 0.4, 0.5, 0.6])
    ba.add_keyframe(depth, rgb, gt_c)
    assert ba.create_surfels_for_keyframe(0, filter_new_surfels=False) > 250000
    worst = 0.0
    for xi in _offsets(0.0005, 0.001):
        init = ob.se3_mul(gt_c, ob.se3_exp(xi))
        est, its, conv = ba.estimate_frame_pose(0, init)
        err = ob.se3_log(ob.se3_mul(ob.se3_inverse(est), gt_c))
        worst = max(worst, np.abs(err).max())
    assert worst <= 8e-5, worst
