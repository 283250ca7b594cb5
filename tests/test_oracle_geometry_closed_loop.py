"""Closed-loop pins of the oracle against the reference's geometry-optimisation test, both solvers.

Restates applications/badslam/src/badslam/test/test_geometry_optimization_geometric_residual.cc:
50-222: one keyframe, per-pixel random depth 1..2 m (scale 1/5000), measured normals forced onto
the viewing ray, surfels created from it, then the depth image is re-noised by up to 5 mm and ten
BundleAdjustment calls (geometry only, 10 iterations each, increase_ba_iteration_count) must
bring every surfel back to within 1e-4 m of the (noisy) depth it projects to: zero failures.
"""
import numpy as np
import pytest

from badslam_amd import se3, synthetic
from oracle import binding as ob

W, H = 640, 480


def _run(use_pcg):
    rng = np.random.Generator(np.random.PCG64(0))
    cam = synthetic.test_camera(W, H)
    s = 1.0 / 5000
    c = ob.make_camera(cam, W, H)
    ba = ob.OracleBA(1000 * 1000, s, 40.0, 1, c, c, use_depth_residuals=True, use_descriptor_residuals=True,
                     min_observation_count=1)
    depth = ((1 + 0.01 * rng.integers(0, 100, size=(H, W))) / s + 0.5).astype(np.uint16)
    depth[0, :] = 65535; depth[-1, :] = 65535; depth[:, 0] = 65535; depth[:, -1] = 65535
    rgb = np.zeros((H, W, 3), np.uint8)
    gt = ob.se3_exp([0.1, 0.2, 0.3, 0.4, 0.5, 0.6])
    ba.add_keyframe(depth, rgb, gt)
    # normals forced to the (negated, normalised) viewing ray of each pixel centre
    fx, fy, cx, cy = [float(v) for v in cam]
    xs = (np.arange(W) - (cx - 0.5)) / fx
    ys = (np.arange(H) - (cy - 0.5)) / fy
    d = np.stack(np.broadcast_arrays(xs[None, :], ys[:, None], np.ones((H, W))), -1)
    n = -d / np.linalg.norm(d, axis=-1, keepdims=True)
    L = ob.lib()
    normals = ba.kf_arrays(0)["normals"]
    for y in range(H):
        for x in range(W):
            normals[y, x] = L.orc_pack_normal8(float(n[y, x, 0]), float(n[y, x, 1]))
    assert ba.create_surfels_for_keyframe(0, filter_new_surfels=False) > 300000
    noisy = depth.astype(np.int64) + ((0.0001 * rng.integers(0, 50, size=(H, W))) / s).astype(np.int64)
    noisy = (noisy & 0xffff).astype(np.uint16)
    ba.kf_arrays(0)["depth"][:] = noisy
    for _ in range(10):
        ba.bundle_adjustment(optimize_poses=False, optimize_geometry=True, min_iterations=10, max_iterations=10,
                             use_pcg=use_pcg, increase_ba_iteration_count=True)
    n_s = ba.surfels_size
    assert n_s > 250000
    P = ba.surfel_data[0:3, :n_s].astype(np.float64)
    F = np.array(list(ba.keyframes[0].frame_T_global), np.float64).reshape(3, 4)
    local = F[:, :3] @ P + F[:, 3:4]
    px = fx * local[0] / local[2] + cx
    py = fy * local[1] / local[2] + cy
    vis = (local[2] > 0) & (px >= 0) & (py >= 0) & (px < W) & (py < H)
    expected = s * noisy[py[vis].astype(int), px[vis].astype(int)]
    err = np.abs(local[2][vis] - expected)
    return int((err > 1e-4).sum()), int(vis.sum())


@pytest.mark.slow
def test_alternating_geometry_optimization_with_geometric_residual():
    fails, n = _run(use_pcg=False)
    assert n > 250000
    assert fails == 0


@pytest.mark.slow
def test_pcg_geometry_optimization_with_geometric_residual():
    fails, n = _run(use_pcg=True)
    assert n > 250000
    assert fails == 0
