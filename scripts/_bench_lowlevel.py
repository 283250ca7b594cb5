#!/usr/bin/env python3
"""bench.py -- BA iterations/s of the MI355X-native direct bundle-adjustment backend.

One "step" = one iteration of the alternating scheme (B/direct_ba_alternating.cc:345-718 of the
reference): surfel activation, geometry optimisation (normals + joint position/descriptor solve)
and pose optimisation of every keyframe, geometric + photometric residuals, fixed surfel set,
full keyframe window.  Workload at N=1: BASELINE.json configs[2] -- synthetic 640x480, 200
keyframes, 3 M surfels.  With --gpus N the surfels are sharded over the ranks (keyframe images
replicated) and the per-keyframe pose normal equations are all-reduced over RCCL (strong
scaling: the same scene, the same result).

Prints ONE JSON line on rank 0.
"""
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


def cpu_baseline(args, log):
    """Times the oracle's full cost evaluation (the reference has no CPU BA path)
    on a bounded sample of the same kind of scene, on this box's host cores."""
    from tests import common
    from oracle import binding as ob
    K, W, H = 8, args.width, args.height
    scene = common.synthetic.make_scene(K, W, H, seed=args.seed, cell=2)
    ba = common.build_oracle(scene, 2000000)
    n = ba.surfels_size
    ba.evaluate_cost()  # warm
    reps, t0 = 0, time.time()
    while True:
        cost, nres = ba.evaluate_cost()
        reps += 1
        if time.time() - t0 > args.cpu_baseline_seconds:
            break
    dt = (time.time() - t0) / reps
    pairs_per_s = K * n / dt
    return dict(pairs_per_s=pairs_per_s, seconds_per_eval=dt, K=K, N=n, cores=ob.lib().orc_num_threads(), nres=nres)


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    from badslam_amd import capi
    ctx, g, data, poses_gt = build_scene(args, log)
    N_total = data.shape[1]
    if args.build_only:
        return
    # surfel sharding: rank r owns a contiguous slice (keyframe images replicated on every rank)
    lo, hi = (N_total * rank) // world, (N_total * (rank + 1)) // world
    g.upload_surfels(data[:, lo:hi], np.zeros(hi - lo, np.uint8))
    hook_keepalive = None
    if world > 1:
        from badslam_amd import multigpu
        hook_keepalive = multigpu.install_allreduce(ctx, dist)
    capi.check(ctx.lib.bahip_set_profiling(ctx.handle, 1))

    stats = dict(rounds=[], gn_steps=[])
    for _ in range(args.warmup):
        ba_iteration(g, stats)
    ctx.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    stats = dict(rounds=[], gn_steps=[])
    stage_ms = np.zeros(4)
    stage_launches = np.zeros(4, dtype=np.int64)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ba_iteration(g, stats)
        for s in range(4):
            ms, n = C.c_float(), C.c_int()
            capi.check(ctx.lib.bahip_last_stage_time_ms(ctx.handle, s, C.byref(ms), C.byref(n)))
            stage_ms[s] += ms.value
            stage_launches[s] += n.value
    ctx.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        K, W, H = args.keyframes, args.width, args.height
        R = float(np.mean(stats["rounds"]))
        Rbar = float(np.sum(stats["gn_steps"])) / (len(stats["gn_steps"]) * K)
        # algorithmic bytes: per pose-accumulate launch = one GN round over all keyframes
        # still iterating; charge the full K (upper bound on compulsory traffic per launch)
        bytes_pose_launch = N_total / world * 28 + K * W * H * 5
        launches = max(1, int(stage_launches[2]))
        avg_ms = stage_ms[2] / launches
        achieved = bytes_pose_launch / (avg_ms * 1e-3) / 1e9
        b_alg_iter = N_total * (17 + 21 + 49 + 28 * R) + K * W * H * (4 + 4 + 5 + 5 * Rbar)
        out = {
            "metric": "BA iterations/sec (and ms/iter) at N keyframes x M surfels, 640x480",
            "value": args.steps / elapsed,
            "unit": "BA iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"synthetic {W}x{H}, {K} keyframes, {N_total} surfels, geometry+photometric alternating BA "
                                   f"(BASELINE configs[2])", "keyframes": K, "surfels": int(N_total), "width": W, "height": H,
                       "parallelism": f"surfel-shard x{world}, RCCL all-reduce of pose H,b" if world > 1 else "single GPU",
                       "pose_gn_rounds_per_iteration": R, "pose_gn_steps_per_keyframe": Rbar},
            "stage_ms_per_iteration": {"surfel_activation": stage_ms[0] / args.steps, "geometry_optimization": stage_ms[1] / args.steps,
                                       "pose_accumulate": stage_ms[2] / args.steps, "pose_solve": stage_ms[3] / args.steps},
            "algorithmic_bytes_per_iteration": b_alg_iter,
            "iteration_fraction_of_hbm_roofline": b_alg_iter / (elapsed / args.steps) / (HBM_PEAK_GBS * 1e9),
            "roofline": {"bound": "hbm", "kernel": "pose_accumulate_kernel<true,true>", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": bytes_pose_launch, "avg_launch_ms": avg_ms, "launches": launches},
        }
        if not args.no_cpu_baseline:
            cb = cpu_baseline(args, log)
            out["cpu_baseline"] = {"value": cb["pairs_per_s"], "unit": "surfel-keyframe pairs/s (full cost evaluation)",
                                   "cores": cb["cores"], "kind": "port",
                                   "sample": f"oracle cost evaluation, {cb['K']} keyframes x {cb['N']} surfels {W}x{H}, "
                                             f"{cb['seconds_per_eval']:.2f} s per evaluation; one BA iteration at the bench "
                                             f"size needs >= {(3 + R):.1f} such sweeps over {K}x{N_total} pairs",
                                   "equivalent_ba_iterations_per_s": cb["pairs_per_s"] / ((3 + R) * K * N_total)}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
