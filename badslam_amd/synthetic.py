"""Procedural RGB-D scenes in the style of the reference's own tests.

The reference has no datasets or fixtures in-tree; every BA test renders random planes
(applications/badslam/src/badslam/test/test_intrinsics_optimization_geometric_residual.cc:109-167,
259-311) textured with a three-sinusoid pattern
(test_intrinsics_optimization_photometric_residual.cc:50-94).  This module restates that generator
with numpy and a portable PRNG (numpy PCG64) -- the reference uses glibc rand() + Eigen::Random,
which are not reproducible elsewhere.  Pure host-side data generation: no oracle, no GPU.
"""
from dataclasses import dataclass, field
from typing import List

import numpy as np

from . import se3


def test_camera(width: int, height: int):
    """{fx, fy, cx, cy} exactly as the reference tests build them (pixel-corner convention),
    test_pose_optimization_geometric_residual.cc:56."""
    return np.array([0.5 * height, 0.5 * height, 0.5 * width - 0.5, 0.5 * height - 0.5], dtype=np.float32)


@dataclass
class Scene:
    width: int
    height: int
    camera: np.ndarray                 # fx, fy, cx, cy (float32), depth == colour camera
    raw_to_float_depth: float
    baseline_fx: float
    cell: int
    planes: np.ndarray                 # (P, 4): unit normal, offset; n.x + d = 0
    poses_gt: List[np.ndarray] = field(default_factory=list)     # global_T_frame, [qx qy qz qw tx ty tz]
    depth: List[np.ndarray] = field(default_factory=list)        # (H, W) uint16 raw depth
    rgb: List[np.ndarray] = field(default_factory=list)          # (H, W, 3) uint8


def random_planes(rng: np.random.Generator, count: int, offset: float = 2.5, slope: float = 1.0) -> np.ndarray:
    """Planes n.x + offset = 0 with n = (slope*u1, slope*u2, -1)/|.|, u uniform in [-1, 1]
    (slope 1 = the reference tests' Vec3f::Random() normals)."""
    n = slope * rng.uniform(-1.0, 1.0, size=(count, 3))
    n[:, 2] = -1.0
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    return np.concatenate([n, np.full((count, 1), offset)], axis=1)


def render_planes(pose, planes, camera, width, height, raw_to_float_depth, textured=True,
                  invalid_border=True):
    """Nearest ray/plane hit per pixel -> (uint16 raw depth, uint8 RGB).
    Ray through the pixel centre (UnprojectFromPixelCenterConv), depth = ray parameter since the
    direction has z = 1 in the camera frame."""
    fx, fy, cx, cy = [float(v) for v in camera]
    R = se3.quat_to_rot(pose[:4])
    o = np.asarray(pose[4:], dtype=np.float64)
    xs = (np.arange(width, dtype=np.float64) - (cx - 0.5)) / fx
    ys = (np.arange(height, dtype=np.float64) - (cy - 0.5)) / fy
    dirs = np.stack(np.broadcast_arrays(xs[None, :], ys[:, None], np.ones((height, width))), axis=-1)  # H,W,3
    gdirs = dirs @ R.T
    best = np.full((height, width), np.inf)
    for pl in planes:
        n, d = pl[:3], pl[3]
        denom = gdirs @ n
        with np.errstate(divide='ignore', invalid='ignore'):
            t = -(n @ o + d) / denom
        ok = (t > 0) & np.isfinite(t) & (t < best)
        best = np.where(ok, t, best)
    hit = np.isfinite(best)
    raw = np.where(hit, np.minimum(65535.0, np.floor(best / raw_to_float_depth + 0.5)), 65535.0).astype(np.uint16)
    if invalid_border:
        raw[0, :] = 65535; raw[-1, :] = 65535; raw[:, 0] = 65535; raw[:, -1] = 65535
    rgb = np.zeros((height, width, 3), dtype=np.uint8)
    if textured:
        z = np.where(hit, best, 0.0)
        gp = o[None, None, :] + gdirs * z[..., None]
        k = 200.0
        def chan(a, b):
            return np.floor((255 / 2.0) * (1.0 + np.sin(0.15 * k * a + 0.5 * np.sin(0.25 * k * b)))).astype(np.uint8)
        rgb[..., 0] = chan(gp[..., 0], gp[..., 1])
        rgb[..., 1] = chan(gp[..., 1], gp[..., 2])
        rgb[..., 2] = chan(gp[..., 2], gp[..., 0])
        rgb[~hit] = 0
    return raw, rgb


def _render_job(job):
    pose, planes, camera, width, height, raw_to_float_depth = job
    return render_planes(pose, planes, camera, width, height, raw_to_float_depth)


def render_many(poses, planes, camera, width, height, raw_to_float_depth, workers=None):
    """render_planes for many poses, in order, on a pool of host processes (a 1280 x 960 frame takes ~1 s of numpy; a
    1000-keyframe scene is minutes on one core).  Spawned workers: the caller may already hold a HIP context, which must not
    be forked.  Yields (raw depth, rgb) per pose."""
    import multiprocessing as mp
    import os
    jobs = [(p, planes, camera, width, height, raw_to_float_depth) for p in poses]
    if workers is None:
        workers = min(len(jobs) // 4, max(1, (os.cpu_count() or 1) - 2), 96)
        if os.environ.get("BADSLAM_RENDER_WORKERS"):      # e.g. under a profiler, which attaches to every spawned worker
            workers = max(1, min(workers, int(os.environ["BADSLAM_RENDER_WORKERS"])))
    if workers <= 1:
        for job in jobs:
            yield _render_job(job)
        return
    with mp.get_context("spawn").Pool(workers) as pool:
        for out in pool.imap(_render_job, jobs, chunksize=1):
            yield out


def make_scene(num_keyframes: int, width: int = 640, height: int = 480, seed: int = 0,
               num_planes: int = 20, raw_to_float_depth: float = 1.0 / 5000, baseline_fx: float = 40.0,
               cell: int = 2, translation_range: float = 3.0, rotation_range: float = 1.4,
               textured: bool = True, plane_distance: float = 2.5) -> Scene:
    """K keyframes scattered around a first pose: T_k = T_0 * exp(xi_k), xi translation uniform in
    +-translation_range/2 and rotation uniform in +-rotation_range/2 (reference:
    test_intrinsics_optimization_geometric_residual.cc:285-297 uses 3.0 m and 1.4 rad)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    cam = test_camera(width, height)
    scene = Scene(width, height, cam, raw_to_float_depth, baseline_fx, cell, random_planes(rng, num_planes, offset=plane_distance))
    T0 = se3.exp([0.01, 0.02, 0.03, 0.004, 0.005, 0.006])
    for _ in range(num_keyframes):
        xi = np.concatenate([translation_range * (rng.random(3) - 0.5), rotation_range * (rng.random(3) - 0.5)])
        T = se3.mul(T0, se3.exp(xi))
        raw, rgb = render_planes(T, scene.planes, cam, width, height, raw_to_float_depth, textured)
        scene.poses_gt.append(T)
        scene.depth.append(raw)
        scene.rgb.append(rgb)
    return scene


def perturb_pose(rng: np.random.Generator, T, sigma_t=0.005, sigma_r=0.001):
    """T * exp(N(0, sigma)) -- magnitudes from test_pose_optimization_geometric_residual.cc:134-135."""
    xi = np.concatenate([rng.normal(0, sigma_t, 3), rng.normal(0, sigma_r, 3)])
    return se3.mul(T, se3.exp(xi))
