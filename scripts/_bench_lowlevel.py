#!/usr/bin/env python3
"""Scene builder and one BA iteration through the low-level C-ABI wrapper (badslam_amd/lowlevel.py): what bench.py was
before it drove the C++ DirectBA.  Kept for scripts/count_pairs.py (work census of one sweep over the bench scene)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--keyframes", type=int, default=200)
    p.add_argument("--surfels", type=int, default=3000000)
    p.add_argument("--width", type=int, default=640)
    p.add_argument("--height", type=int, default=480)
    p.add_argument("--seed", type=int, default=0)
    # Scene extent: ~20 gently sloped planes about 2.5 m in front of a camera rig that moves laterally.
    # Sized so that, at cell 2, all keyframes together create ~3.0 M surfels and every surfel is seen
    # by ~5 keyframes (each keyframe sees ~2.5 % of the surfels).
    p.add_argument("--extent-x", type=float, default=17.3, help="keyframe x positions uniform in +-extent/2 [m]")
    p.add_argument("--extent-y", type=float, default=13.0)
    p.add_argument("--extent-z", type=float, default=0.6)
    p.add_argument("--rotation-range", type=float, default=0.6, help="rotation vector components uniform in +-range/2 [rad]")
    p.add_argument("--plane-slope", type=float, default=0.05)
    p.add_argument("--cell", type=int, default=2, help="sparse_surfel_cell_size")
    p.add_argument("--build-only", action="store_true")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-baseline-seconds", type=float, default=15.0)
    return p.parse_args()


def build_scene(args, log):
    """Synthetic planes scene: K keyframes rendered on the host, preprocessed and
    turned into surfels by the HIP path itself; poses and surfels perturbed before timing."""
    from badslam_amd import lowlevel as ll, synthetic
    t0 = time.time()
    rng = np.random.Generator(np.random.PCG64(args.seed))
    cam = synthetic.test_camera(args.width, args.height)
    planes = synthetic.random_planes(rng, 20, slope=args.plane_slope)
    from badslam_amd import se3
    T0 = se3.exp([0.01, 0.02, 0.03, 0.004, 0.005, 0.006])
    ext = np.array([args.extent_x, args.extent_y, args.extent_z])
    # small scenes (parity configs) shrink the extent with the keyframe count so coverage stays ~5x
    ext[:2] *= np.sqrt(args.keyframes / 200.0 * (args.width * args.height) / (640.0 * 480.0))
    ctx = ll.Context()
    cells = ((args.width - 1) // args.cell + 1) * ((args.height - 1) // args.cell + 1)
    cap = args.keyframes * cells + 1024   # worst case: no overlap between keyframes (288 GB of HBM: not a concern)
    g = ll.Scene(ctx, cap, 1.0 / 5000, 40.0, args.cell, ll.make_camera(cam, args.width, args.height),
                 ll.make_camera(cam, args.width, args.height))
    poses_gt = []
    for k in range(args.keyframes):
        xi = np.concatenate([ext * (rng.random(3) - 0.5), args.rotation_range * (rng.random(3) - 0.5)])
        T = se3.mul(T0, se3.exp(xi))
        raw, rgb = synthetic.render_planes(T, planes, cam, args.width, args.height, 1.0 / 5000)
        g.add_keyframe(raw, rgb, T)
        poses_gt.append(T)
    log(f"rendered + preprocessed {args.keyframes} keyframes in {time.time() - t0:.1f}s")
    t1 = time.time()
    # surfels from the keyframes (unfiltered creation, reference B/direct_ba.cc:340-405), cycling with
    # decreasing sparsity until the target count is reached
    per_kf = []
    for k in range(args.keyframes):
        per_kf.append(g.create_surfels_for_keyframe(k, filter_new_surfels=False))
    created = g.surfels_size
    if g.surfels_size > args.surfels:
        g.surfels_size = g.surfel_count = args.surfels
    log(f"created {created} surfels from {args.keyframes} keyframes (min/median/max per keyframe "
        f"{min(per_kf)}/{int(np.median(per_kf))}/{max(per_kf)}) in {time.time() - t1:.1f}s; using {g.surfels_size}")
    # perturbation: poses * exp(N(0, 5 mm / 1 mrad)); surfels + U(0, 5 mm) along z
    prng = np.random.Generator(np.random.PCG64(args.seed + 1))
    for k in range(args.keyframes):
        g.keyframes[k]["pose"] = np.asarray(synthetic.perturb_pose(prng, poses_gt[k]), np.float32)
    data = g.download_surfels()
    data[2] += prng.uniform(0, 0.005, data.shape[1]).astype(np.float32)
    return ctx, g, data, poses_gt


def ba_iteration(g, stats):
    """One alternating-scheme iteration on the bound scene (fixed keyframe window: every keyframe
    is kActive at the start of each iteration, B/direct_ba_alternating.cc:354-372)."""
    from badslam_amd import capi
    for kf in g.keyframes:
        kf["activation"] = capi.KF_ACTIVE
    g.bind_keyframes()
    g.update_surfel_activation()
    g.optimize_geometry_iteration(True, True)
    poses, its, conv, rounds = g.estimate_keyframe_poses(True, True)
    for k, kf in enumerate(g.keyframes):
        kf["pose"] = poses[k].astype(np.float32)
    stats["rounds"].append(rounds)
    stats["gn_steps"].append(int(its.sum()))
