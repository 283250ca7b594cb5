#!/usr/bin/env python3
"""bench.py -- BA iterations/s of the MI355X-native direct bundle-adjustment backend.

One "step" = one iteration of the alternating scheme (B/direct_ba_alternating.cc:345-718 of the
reference): surfel activation, geometry optimisation (normals + joint position/descriptor solve)
and pose optimisation of every keyframe, geometric + photometric residuals, fixed surfel set,
full keyframe window.  The timed region is ONE call of the C++ host class
`vis::DirectBA::BundleAdjustment(min_iterations = max_iterations = steps)` -- the same call the
reference's BadSlam::RunBundleAdjustment makes (B/bad_slam.cc:780) -- through the flat C view
include/badslam_directba.h; Python only prepares the synthetic scene and prints the result.

Workload at N=1: BASELINE.json configs[2] -- synthetic 640x480, 200 keyframes, 3 M surfels.  With
--gpus N the surfels are sharded over the ranks (keyframe images replicated) and the per-keyframe
pose normal equations are all-reduced over RCCL (strong scaling: the same scene, the same result).

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
FP32_VECTOR_PEAK_TFLOPS = 157.3   # same guide: peak FP32 (vector)
SURFEL_ROWS = 17        # BAHIP_SURFEL_ATTRIBUTE_COUNT
STAGES = ("surfel_activation", "geometry_optimization", "pose_accumulate", "pose_solve", "intrinsics_optimization")


# The flavour `value` is measured with.  "fast" is the arithmetic of the reference's own build (nvcc -use_fast_math); the exact flavour
# (every bit the CPU oracle's) is what the parity tests hold and is measured beside it (`exact_arithmetic` / `fast_math` in the line).
DEFAULT_ARITHMETIC = "fast"


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)     # 20 timed iterations after 3 warm-up
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--keyframes", type=int, default=200)
    p.add_argument("--surfels", type=int, default=3000000)
    p.add_argument("--width", type=int, default=640)
    p.add_argument("--height", type=int, default=480)
    p.add_argument("--seed", type=int, default=0)
    # Scene extent: ~20 gently sloped planes about 2.5 m in front of a camera rig that moves laterally.
    # Sized so that, at cell 2, all keyframes together create ~3.0 M surfels (DESIGN.md "bench scene").
    p.add_argument("--extent-x", type=float, default=17.3, help="keyframe x positions uniform in +-extent/2 [m]")
    p.add_argument("--extent-y", type=float, default=13.0)
    p.add_argument("--extent-z", type=float, default=0.6)
    p.add_argument("--rotation-range", type=float, default=0.6, help="rotation vector components uniform in +-range/2 [rad]")
    p.add_argument("--plane-slope", type=float, default=0.05)
    p.add_argument("--cell", type=int, default=2, help="sparse_surfel_cell_size")
    p.add_argument("--pcg", action="store_true", help="time the PCG scheme instead of the alternating one (single GPU)")
    p.add_argument("--intrinsics", action="store_true",
                   help="also optimise the depth and colour intrinsics in every iteration (BASELINE configs[4]: joint BA)")
    p.add_argument("--force-allreduce", action="store_true",
                   help="install the RCCL all-reduce hook even with one rank (measures the cost of the exchange path)")
    p.add_argument("--emulate-world", type=int, default=0,
                   help="single process: time rank 0's share of an N-rank run (with --force-allreduce: plus the exchange path); "
                        "a planning aid, the JSON line then describes that share, not the whole job")
    p.add_argument("--emulate-rank", type=int, default=0)
    p.add_argument("--shard-chunk", type=int, default=4096, help="surfels per chunk of the chunk-cyclic partition; 0 = contiguous")
    p.add_argument("--shard", choices=["surfels", "keyframes"], default="surfels",
                   help="what N > 1 ranks divide: surfels (keyframe images replicated; the production axis, any N) or keyframes "
                        "(BASELINE configs[3] as written: every rank holds all surfels and the images of the keyframes k with "
                        "k %% N == rank; N = 2, 4 or 8 (8: the 8-class definition of the per-surfel sums); class partials of the geometry step and the pose normal equations are "
                        "exchanged; same bits as one GPU)")
    p.add_argument("--no-spatial-sort", action="store_true", help="leave the surfels in creation order")
    p.add_argument("--sort-cell", type=float, default=0.02, help="grid cell of DirectBA::SortSurfelsSpatially [m] (its default: 0.02)")
    p.add_argument("--no-prepass", action="store_true", help="do not run the iterations once before the warm-up and the timed region (see PRE-PASS in main)")
    p.add_argument("--launch-shapes", default="", help="experiment: 'tile_waves,pose_parts' forced through bahip_debug_set_launch_shapes (0 = heuristic)")
    p.add_argument("--arithmetic", choices=["exact", "fast"], default=os.environ.get("BENCH_ARITHMETIC", DEFAULT_ARITHMETIC),
                   help="arithmetic flavour of the sweeps in the timed region (bahip_context_set_arithmetic); the other flavour is measured "
                        "after it, by the same protocol, and reported beside `value`")
    p.add_argument("--build-only", action="store_true")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-extras", action="store_true",
                   help="skip the untimed extra measurements after the timed region (intrinsics stage, PCG scheme)")
    return p.parse_args()


def build_scene(args, log):
    """Synthetic planes scene: K keyframes rendered on the host, preprocessed and
    turned into surfels by the HIP path itself (Keyframe ctor + DirectBA::CreateSurfelsForKeyframe);
    poses and surfels perturbed before timing."""
    from badslam_amd import se3, synthetic
    from badslam_amd.directba import DirectBA
    t0 = time.time()
    rng = np.random.Generator(np.random.PCG64(args.seed))
    cam = synthetic.test_camera(args.width, args.height)
    T0 = se3.exp([0.01, 0.02, 0.03, 0.004, 0.005, 0.006])
    ext = np.array([args.extent_x, args.extent_y, args.extent_z])
    cells = ((args.width - 1) // args.cell + 1) * ((args.height - 1) // args.cell + 1)
    # The area the keyframes spread over follows the surfel count asked for: a frame shows the same piece of the world at every
    # resolution, and the surfels it creates are one per sparse cell of the part nobody covered yet, so
    # (surfels created) ~ area x cells per frame.  The defaults (configs[2]: 3 M surfels, 320 x 240 cells) are scale 1 and
    # yield 3.02 M; configs[4] (20 M, 640 x 480 cells, 1000 keyframes) gets 1.67 x that area and 5 x the keyframes, i.e. every
    # surfel in ~50 keyframes.  "All the surfels the keyframes create" (surfels >= 1e8, the parity slices) keeps the
    # observations per surfel of the defaults instead: area ~ keyframes.
    scale2 = (args.surfels / 3.0e6) * (76800.0 / cells) if args.surfels < 10 ** 8 else args.keyframes / 200.0
    ext[:2] *= np.sqrt(scale2)
    planes = synthetic.random_planes(rng, 20, slope=args.plane_slope)
    cap = args.keyframes * cells + 1024   # worst case: no overlap between keyframes (288 GB of HBM: not a concern)
    if args.launch_shapes:
        from badslam_amd import capi
        tile_waves, pose_parts = (int(v) for v in args.launch_shapes.split(","))
        capi.load().bahip_debug_set_launch_shapes(tile_waves, pose_parts)
    ba = DirectBA(cap, 1.0 / 5000, 40.0, args.cell, args.width, args.height, cam, cam)
    poses_gt = []
    for k in range(args.keyframes):
        xi = np.concatenate([ext * (rng.random(3) - 0.5), args.rotation_range * (rng.random(3) - 0.5)])
        if scale2 <= 1.0:
            poses_gt.append(se3.mul(T0, se3.exp(xi)))
        else:
            # exp() of a twist moves the camera by ~ 1/2 omega x t along its axis as well: 1.3 m at the corners of the default
            # area (part of that scene's depth variety), but on a wider area it carries the corner cameras into the walls
            # (depth 0.04 .. 0.2 m; those poses never settle).  There: translation, then rotation.
            poses_gt.append(se3.mul(T0, se3.mul(se3.exp(np.concatenate([xi[:3], np.zeros(3)])), se3.exp(np.concatenate([np.zeros(3), xi[3:]])))))
    # rendered on a pool of host processes (the same frames in the same order whatever the pool size), preprocessed on the GPU
    workers = None if args.keyframes * args.width * args.height >= 64 * 640 * 480 else 1
    keep_frames = [] if os.environ.get("BENCH_KEEP_FRAMES") else None   # (tests that build further instances of the same scene)
    for T, (raw, rgb) in zip(poses_gt, synthetic.render_many(poses_gt, planes, cam, args.width, args.height, 1.0 / 5000, workers)):
        ba.AddKeyframe(raw, rgb, T)
        if keep_frames is not None:
            keep_frames.append((raw, rgb))
    build_scene.frames = keep_frames
    log(f"rendered + preprocessed {args.keyframes} keyframes in {time.time() - t0:.1f}s")
    t1 = time.time()
    # surfels from the keyframes (unfiltered creation, reference B/direct_ba.cc:340-405)
    per_kf, before = [], 0
    for k in range(args.keyframes):
        ba.CreateSurfelsForKeyframe(k, filter_new_surfels=False)
        per_kf.append(ba.surfels_size() - before)
        before = ba.surfels_size()
    created = ba.surfels_size()
    if created > 1.02 * args.surfels:
        # far more than asked for (the estimate above is tuned at the defaults): an even subset, so that every keyframe keeps
        # its share of surfels -- a prefix would leave the last keyframes with next to nothing to estimate their poses from
        everything = ba.download_surfels(rows=SURFEL_ROWS)
        keep = (np.arange(args.surfels, dtype=np.int64) * created) // args.surfels
        ba.upload_surfels(np.ascontiguousarray(everything[:, keep]))
        del everything
    elif created > args.surfels:
        ba.SetSurfelCount(args.surfels, args.surfels)
    creation_order = None
    if not args.no_spatial_sort:
        if not args.no_extras:
            creation_order = ba.download_surfels(rows=SURFEL_ROWS)   # for `unsorted_ba_iterations_per_s`
        # The order a caller of the reference's API ends up with: DirectBA::PerformBASchemeEndTasks re-establishes the Morton
        # order of the buffer whenever surfels were appended or moved (round 4; direct_ba.h: SetSpatialSortCellSize), and every
        # BundleAdjustment(increase_ba_iteration_count = true) runs those end tasks.  Scene construction calls the same
        # operation directly, outside the timed region.
        ba.SortSurfelsSpatially(args.sort_cell)
    log(f"created {created} surfels from {args.keyframes} keyframes (min/median/max per keyframe "
        f"{min(per_kf)}/{int(np.median(per_kf))}/{max(per_kf)}) in {time.time() - t1:.1f}s; using {ba.surfels_size()}")
    # perturbation: poses * exp(N(0, 5 mm / 1 mrad)); surfels + U(0, 5 mm) along z
    prng = np.random.Generator(np.random.PCG64(args.seed + 1))
    for k in range(args.keyframes):
        ba.set_keyframe_pose(k, synthetic.perturb_pose(prng, poses_gt[k]))
    data = ba.download_surfels(rows=SURFEL_ROWS)
    data[2] += prng.uniform(0, 0.005, data.shape[1]).astype(np.float32)
    if creation_order is not None:
        creation_order[2] += prng.uniform(0, 0.005, creation_order.shape[1]).astype(np.float32)
    build_scene.creation_order = creation_order
    return ba, data, poses_gt


def committed_profile(args, intrinsics=None, pcg=None):
    """The newest committed PMC summary (scripts/profile_round.sh -> profiles/<tag>_pmc_per_kernel.json) whose recorded
    config is THIS run's workload -- counters of another scene are not this run's traffic.  None if there is none.
    intrinsics / pcg: the leg of the extras (the same scene with the intrinsics step / the PCG scheme) instead of the run's own."""
    import glob
    from badslam_amd import buildinfo
    want = {"keyframes": args.keyframes, "surfels": args.surfels, "width": args.width, "height": args.height,
            "intrinsics": bool(args.intrinsics if intrinsics is None else intrinsics), "pcg": bool(args.pcg if pcg is None else pcg),
            "arithmetic": args.arithmetic}
    digest = buildinfo.csrc_digest()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_per_kernel.json")), reverse=True):
        with open(path) as f:
            pmc = json.load(f)
        config = dict(pmc.get("config") or {})
        config.setdefault("arithmetic", "exact")          # (profiles older than the flavours)
        if config != want:
            continue
        # Counters of OTHER kernels are not this run's traffic: a profile is quoted only while the kernel sources are the ones it was
        # taken on.  A stale one is named, so that the line says why `traffic` is null.
        if pmc.get("csrc_digest") != digest:
            STALE_PROFILES.append(os.path.relpath(path, ROOT))
            continue
        return pmc, os.path.relpath(path, ROOT)
    return None, None


STALE_PROFILES = []


def pmc_kernel_entry(pmc, source, kernel_prefix):
    """HBM-side bytes per launch of one kernel from the PMC passes (FETCH_SIZE and WRITE_SIZE, collected in separate passes,
    in KB).  FETCH_SIZE counts fabric read requests; MI355X_MICROARCH.md "HBM" calibrates it for wide coalesced reads
    (128-byte requests tallied at 64 bytes -> x2).  These sweeps issue 4-byte gathers, so the factor used here is the one
    measured on THIS access pattern by the calibration kernel of the same profile run (pmc["fetch_calibration"]: a gather of
    known size, bytes actually requested / bytes FETCH_SIZE reported), not the guide's 2.0; without it the entry is None."""
    if pmc is None:
        return None
    cal = pmc.get("fetch_calibration")
    for name, counters in pmc.get("kernels", {}).items():
        if name.startswith(kernel_prefix) and "FETCH_SIZE" in counters and cal:
            fetch = counters["FETCH_SIZE"]["avg_per_launch"] * 1024.0 * cal["factor"]
            write = counters.get("WRITE_SIZE", {"avg_per_launch": 0.0})["avg_per_launch"] * 1024.0
            out = {"bytes": fetch + write, "fetch_bytes": fetch, "write_bytes": write, "fetch_factor": cal["factor"], "source": source}
            for key in ("valu_issue_fraction", "valu_cycles_per_instruction", "fp32_flops_per_launch", "non_arithmetic_valu_fraction",
                        "valu_fraction_int32", "valu_fraction_int64", "valu_fraction_cvt"):
                if key in counters:
                    out[key] = counters[key]
            return out
    return None


def cpu_baseline(args, ba, data, log):
    """One full cost evaluation -- every residual of every (surfel, keyframe) pair, no Jacobians: the reference has no CPU BA
    path, its only CPU-side notion of the cost is this sum -- of THE BENCH SCENE on this box's host
    cores: the very keyframe images the GPU path worked on (downloaded) and the same surfels, evaluated by the reference's own
    functions (oracle/_ref, kind "reference") or, where that library is absent, by the oracle's restatement (kind "port").
    Nothing is extrapolated."""
    from badslam_amd import synthetic
    from oracle import binding as ob
    t0 = time.time()
    cam = synthetic.test_camera(args.width, args.height)
    K, N = ba.keyframe_count(), data.shape[1]
    images = {name: np.stack([ba.keyframe_image(k, name) for k in range(K)]) for name in ("depth", "normals", "radius", "color")}
    poses = np.stack([ba.keyframe_pose(k) for k in range(K)])
    from oracle import ref_binding as rb
    if os.path.exists(rb.LIB_PATH) and not os.environ.get("BENCH_CPU_BASELINE_PORT"):
        # the REFERENCE's own functions (its device-math headers compiled for the host: oracle/_ref, built in the build container
        # from /root/reference and shipped prebuilt), OpenMP over the surfels, one keyframe per call -- in a process of its own,
        # which holds neither the HIP runtime nor torch (in this process, next to them, the library crashed on the GPU box)
        import subprocess
        import tempfile
        with tempfile.TemporaryDirectory(prefix="bench_cpu_baseline_") as d:
            for name, a in images.items():
                np.save(os.path.join(d, name + ".npy"), a)
            np.save(os.path.join(d, "poses.npy"), poses)
            np.save(os.path.join(d, "surfels.npy"), np.ascontiguousarray(data))
            with open(os.path.join(d, "meta.json"), "w") as f:
                json.dump(dict(width=args.width, height=args.height, keyframes=K, surfels=N, camera=[float(v) for v in cam],
                               raw_to_float_depth=1.0 / 5000, baseline_fx=40.0, cell=args.cell), f)
            log(f"cpu baseline: scene written for the reference's functions in {time.time() - t0:.1f}s ({K} keyframes, {N} surfels)")
            proc = subprocess.run([sys.executable, "-m", "oracle.ref_cost_worker", d], capture_output=True, text=True, cwd=ROOT, timeout=900)
        if proc.returncode == 0:
            r = json.loads(proc.stdout.strip().splitlines()[-1])
            reference = dict(pairs_per_s=K * N / r["seconds"], seconds_per_eval=r["seconds"], K=K, N=N, cores=r["cores"], nres=r["nres"], cost=r["cost"],
                             kind="reference",
                             who="the reference's own association / residual / robust-cost functions (oracle/_ref: B/surfel_projection_nvcc_only.cuh, "
                                 "B/cost_function.cuh, B/robust_weighting.cuh compiled for the host, OpenMP over the surfels, in a process of its own)")
        else:
            log(f"cpu baseline: the reference library failed (exit {proc.returncode}: {proc.stderr.strip()[-300:]}); timing the oracle's restatement only")
            reference = None
    else:
        reference = None
    orc = ob.OracleBA(N + 64, 1.0 / 5000, 40.0, args.cell, ob.make_camera(cam, args.width, args.height),
                      ob.make_camera(cam, args.width, args.height))
    for k in range(K):
        orc.add_preprocessed_keyframe(images["depth"][k], images["normals"][k], images["radius"][k], images["color"][k], poses[k])
    orc.surfel_data[:data.shape[0], :N] = data
    orc.surfels.surfels_size = orc.surfels.surfel_count = N
    log(f"cpu baseline: scene handed to the oracle in {time.time() - t0:.1f}s ({K} keyframes, {N} surfels)")
    cores = ob.lib().orc_num_threads()
    t1 = time.time()
    cost, nres = orc.evaluate_cost()
    dt = time.time() - t1
    port = dict(pairs_per_s=K * N / dt, seconds_per_eval=dt, K=K, N=N, cores=cores, nres=nres, cost=cost, kind="port",
                who="the oracle's restatement (OpenMP over the surfels, keyframes culled per 64-surfel tile against their frusta like the GPU sweeps)")
    # both legs every time: the reference's functions are a brute-force K x N loop, the port culls -- a line that
    # switched from one to the other moved the "GPU / CPU" ratio 45-fold for no reason
    if reference is not None:
        reference["port"] = port
        return reference
    return port


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks here, one per GPU, exactly the way the
    driver does it (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`), and pass
    rank 0's JSON line through.  Fails loudly when the box has fewer than N devices (BENCH_DIST_BACKEND=gloo lets ranks share
    a device: a plumbing test, not a measurement)."""
    import socket
    import subprocess
    from badslam_amd import capi
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    devices = int(capi.load().bahip_device_count())
    if devices < 1 or (backend == "nccl" and devices < args.gpus):
        print(f"bench.py: --gpus {args.gpus} needs {args.gpus} HIP devices, this box has {devices} "
              f"(one rank per GPU over RCCL; nothing is measured on fewer)", file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        sys.exit(2)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        # a line that says "n_gpus": 1 for a run that was asked to use 8 would be read as a scaling result
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ.get('WORLD_SIZE')} rank(s)", file=sys.stderr)
        sys.exit(2)
    # stdout carries exactly one JSON line: anything a library prints to fd 1 (RCCL prints a version banner there)
    # is sent to stderr instead, and the result is written to the original stdout at the end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None

    class Watchdog:
        """A multi-rank job that cannot complete its rendezvous or its first collective hangs without a word (the driver then only
        sees its own time limit).  This prints which step did not finish, on which rank, and ends the rank with a non-zero code."""
        def __init__(self, what, seconds):
            import threading
            self.timer = threading.Timer(seconds, self.fire, (what, seconds))
            self.timer.daemon = True
        def fire(self, what, seconds):
            print(f"bench.py: rank {rank} of {world}: {what} did not finish within {seconds:.0f} s -- not every rank arrived, or the "
                  f"transport (RCCL over xGMI; MASTER_ADDR={os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}) did not come up. "
                  f"NCCL_DEBUG=INFO shows the communicator's rings; BENCH_COLLECTIVE_TIMEOUT_S changes this limit.", file=sys.stderr, flush=True)
            os._exit(4)
        def __enter__(self):
            self.timer.start()
            return self
        def __exit__(self, *exc):
            self.timer.cancel()
            return False

    collective_timeout = float(os.environ.get("BENCH_COLLECTIVE_TIMEOUT_S", "180"))
    if world > 1 or args.force_allreduce:
        import datetime
        import torch.distributed as dist
        # one rank per GPU; BENCH_DIST_BACKEND=gloo lets several ranks share a device (plumbing test on a 1-GPU box: RCCL
        # refuses two ranks on one device, gloo stages the all-reduce through the host)
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
        if backend == "nccl" and torch.cuda.device_count() < world:
            print(f"bench.py: {world} ranks over RCCL need {world} devices, this box has {torch.cuda.device_count()}", file=sys.stderr)
            sys.exit(2)
        device_index = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
        torch.cuda.set_device(device_index)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        with Watchdog("the torch.distributed rendezvous (init_process_group)", collective_timeout):
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index),
                                        timeout=datetime.timedelta(seconds=collective_timeout))
            else:
                dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=collective_timeout))
    else:
        torch.cuda.set_device(0)

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    from badslam_amd import capi, multigpu
    ba, data, poses_gt = build_scene(args, log)
    N_total = data.shape[1]
    start_poses = [ba.keyframe_pose(k) for k in range(args.keyframes)]   # the perturbed poses and the cameras (for the cold-start figure)
    start_cameras = ba.cameras()
    if args.build_only:
        return
    # surfel sharding: rank r owns every world-th chunk of 4096 surfels (keyframe images replicated on every rank)
    shard_world = args.emulate_world if (args.emulate_world > 0 and world == 1) else world
    shard_rank = args.emulate_rank if shard_world != world else rank
    if args.shard_chunk > 0:
        mine = multigpu.shard_chunks(N_total, shard_rank, shard_world, args.shard_chunk)
    else:
        lo, hi = multigpu.shard_range(N_total, shard_rank, shard_world)
        mine = np.arange(lo, hi, dtype=np.int64)
    by_keyframes = args.shard == "keyframes" and shard_world > 1
    if by_keyframes:
        if shard_world not in (2, 4, 8) or args.intrinsics or args.pcg:
            print("bench.py: --shard keyframes takes 2, 4 or 8 ranks and the alternating scheme over poses and geometry (a rank holds "
                  "whole keyframe classes of the per-surfel sums; intrinsics / PCG need surfel sharding)", file=sys.stderr)
            sys.exit(2)
        mine = np.arange(N_total, dtype=np.int64)
    ba.upload_surfels(np.ascontiguousarray(data[:, mine]) if (shard_world > 1 and not by_keyframes) else data)
    ctx = ba.backend_context()
    ba.SetFastArithmetic(args.arithmetic == "fast")
    hook_keepalive = None
    ranks_seen = 1
    if dist is not None:
        with Watchdog("setting up the backend's transport (ncclCommInitRank / the first all-reduce)", collective_timeout):
            if dist.get_backend() == "nccl" and not os.environ.get("BENCH_ALLREDUCE_HOOK"):
                # native path: the backend owns an RCCL communicator and issues ncclAllReduce on its own stream.  Should the
                # communicator not come up on some system (all ranks must agree, hence the vote), the hook over torch.distributed
                # carries the same sums.
                try:
                    multigpu.init_rccl(ctx, dist)
                    native_ok = 1
                except Exception as e:   # noqa: BLE001 -- whatever it is, the run should still be measured
                    log(f"native RCCL path unavailable ({e}); using the torch.distributed hook")
                    native_ok = 0
                vote = torch.tensor([native_ok], device="cuda", dtype=torch.int32)
                dist.all_reduce(vote, op=dist.ReduceOp.MIN)
                if int(vote.item()) == 0:
                    hook_keepalive = multigpu.install_allreduce(ctx, dist)
            else:
                hook_keepalive = multigpu.install_allreduce(ctx, dist)
            # the first exchange of the run, as a probe with a time limit: every rank contributes 1 through the transport the BA
            # loop is going to use; the line reports how many took part
            seen = C.c_int()
            capi.check(ctx.lib.bahip_context_count_ranks(ctx.handle, int(1e3 * collective_timeout), C.byref(seen)))
            ranks_seen = int(seen.value)
            if ranks_seen != world:
                print(f"bench.py: rank {rank}: the probe exchange counted {ranks_seen} rank(s), the launcher started {world}", file=sys.stderr)
                sys.exit(2)
    K = args.keyframes
    if by_keyframes:
        if shard_world == 8:
            ba.SetSumClasses(8)      # eight ranks hold whole classes of the 8-class definition of the per-surfel sums
        ba.SetKeyframeSharding(shard_rank, shard_world)

    def run(iterations, intrinsics=args.intrinsics, pcg=args.pcg):
        # BA iteration counters equal -> BundleAdjustment skips PerformBASchemeEndTasks (fixed surfel set)
        ba.set_ba_iteration_counts(1, 1)
        done, _ = ba.BundleAdjustment(optimize_depth_intrinsics=intrinsics, optimize_color_intrinsics=intrinsics,
                                      do_surfel_updates=False,
                                      optimize_poses=True, optimize_geometry=True, min_iterations=iterations,
                                      max_iterations=iterations, use_pcg=pcg, active_keyframe_window_start=0,
                                      active_keyframe_window_end=K - 1, increase_ba_iteration_count=False)
        assert done == iterations, (done, iterations)

    def reset_to_start_state():
        # back to the perturbed start state: surfels re-uploaded, poses, cameras and cfactor image reset
        ba.upload_surfels(np.ascontiguousarray(data[:, mine]) if (shard_world > 1 and not by_keyframes) else data)
        for k, T in enumerate(start_poses):
            ba.set_keyframe_pose(k, T)
        ba.set_cameras(*start_cameras)
        ba.L.dba_clear_cfactor(ba.h, ba.stream)

    # PRE-PASS (round 5): the iterations of warm-up + timed region run once before, and the scene is reset.  Without it the timed region is
    # the first sustained load of the process and reads 1.5-2 % below its own repeat (alternating runs on one box: 641 / 651 it/s; the
    # geometry launches of a first pass go 786, 771, 767, 756 ... 726 us while those of a second pass over the same work start at 751:
    # clocks and first touch, not the work): `value` is meant to be the rate of a busy GPU.  --no-prepass / BENCH_PREPASS=0 switch it off.
    # (round 6, ADVICE r5: every mode takes it -- PCG and intrinsics runs too --, and the pre-pass itself is timed the way the contract
    # times a run without it: W untimed iterations, then K timed ones; the line carries that figure beside `value`.)
    prepass = (not args.no_prepass) and os.environ.get("BENCH_PREPASS", "1") != "0"
    first_pass_rate = None
    if prepass:
        if args.warmup > 0:
            run(args.warmup)
        ctx.synchronize()
        torch.cuda.synchronize()
        t_first = time.perf_counter()
        run(args.steps)
        ctx.synchronize()
        first_pass_rate = args.steps / (time.perf_counter() - t_first)
        reset_to_start_state()
    if args.warmup > 0:
        run(args.warmup)

    def pose_dispatches():
        n = C.c_longlong()
        capi.check(ctx.lib.bahip_debug_pose_kernel_dispatches(C.byref(n)))
        return int(n.value)

    # THE TIMED REGION: exactly `steps` iterations, no measurement code in it (round 5 measured what the hipEvent pairs round 4 kept
    # inside it cost: 0.8-0.9 % in alternating runs -- every record is a barrier packet between two launches of the device-driven loop).
    capi.check(ctx.lib.bahip_set_profiling(ctx.handle, 0))
    capi.check(ctx.lib.bahip_exchange_stats(ctx.handle, None, None, 1))
    ctx.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    loop_handled0, loop_declined0 = C.c_longlong(), C.c_longlong()
    capi.check(ctx.lib.bahip_debug_alternating_loop_calls(C.byref(loop_handled0), C.byref(loop_declined0)))
    t0 = time.perf_counter()
    run(args.steps)
    ctx.synchronize()
    torch.cuda.synchronize()
    local_elapsed = time.perf_counter() - t0       # this rank alone, before it waits for the others
    loop_handled1, loop_declined1 = C.c_longlong(), C.c_longlong()
    capi.check(ctx.lib.bahip_debug_alternating_loop_calls(C.byref(loop_handled1), C.byref(loop_declined1)))
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    exchange_calls, exchange_bytes = C.c_longlong(), C.c_longlong()
    capi.check(ctx.lib.bahip_exchange_stats(ctx.handle, C.byref(exchange_calls), C.byref(exchange_bytes), 0))
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stats = ba.last_stats()
    timed_end_poses = np.array([ba.keyframe_pose(k) for k in range(args.keyframes)])   # (after the clock has stopped: for the flavours' parity figures)

    def read_stage_timers():
        out_ms, out_n = np.zeros(8), np.zeros(8, dtype=np.int64)
        for s in range(8):
            ms, n = C.c_float(), C.c_int()
            capi.check(ctx.lib.bahip_last_stage_time_ms(ctx.handle, s, C.byref(ms), C.byref(n)))
            out_ms[s], out_n[s] = ms.value, n.value
        return out_ms, out_n

    # THE SAME ITERATIONS ONCE MORE, INSTRUMENTED: the scene is put back to its perturbed start (surfels re-uploaded, poses, cameras
    # and cfactor image reset) and warmed up again -- the backend is deterministic, so iterations warmup + 1 .. warmup + steps repeat
    # bit for bit, the same keyframes taking the same Gauss-Newton steps -- now with a hipEvent pair around every launch of the dominant
    # kernel, on the backend's own stream: the roofline's launch duration, measured live on the work of the timed region.  The line
    # carries this region's own ms per step beside `ms_per_step`.  The other stages are timed in a few extra iterations afterwards.
    reset_to_start_state()
    capi.check(ctx.lib.bahip_set_profiling(ctx.handle, 0))
    if args.warmup > 0:
        run(args.warmup)
    capi.check(ctx.lib.bahip_set_profiling(ctx.handle, 3))
    capi.check(ctx.lib.bahip_debug_pose_form_launches(None, None, 1))
    dispatches_before = pose_dispatches()
    ctx.synchronize()
    t_instr = time.perf_counter()
    run(args.steps)
    ctx.synchronize()
    instrumented_elapsed = time.perf_counter() - t_instr
    dispatches_timed = pose_dispatches() - dispatches_before
    repeat_stats = ba.last_stats()
    if (repeat_stats["pose_rounds"], repeat_stats["pose_steps"]) != (stats["pose_rounds"], stats["pose_steps"]):
        log(f"note: the instrumented repeat took {repeat_stats['pose_rounds']} rounds / {repeat_stats['pose_steps']} steps, the timed region "
            f"{stats['pose_rounds']} / {stats['pose_steps']}")
    stage_ms, stage_launches = read_stage_timers()             # stage 2 over the instrumented region
    form_global, form_lds = C.c_longlong(), C.c_longlong()
    capi.check(ctx.lib.bahip_debug_pose_form_launches(C.byref(form_global), C.byref(form_lds), 0))
    units = C.c_longlong()
    capi.check(ctx.lib.bahip_stage_work_units(ctx.handle, 2, C.byref(units)))
    BREAKDOWN_STEPS = 5                                         # untimed: per-stage breakdown of the iterations that follow
    capi.check(ctx.lib.bahip_set_profiling(ctx.handle, 2))
    run(BREAKDOWN_STEPS)
    ctx.synchronize()
    breakdown_ms, _ = read_stage_timers()

    # THE OTHER ARITHMETIC FLAVOUR, by the same protocol: scene reset to the same start state, the same warm-up, the same
    # number of timed iterations, then the instrumented repeat for the pose sweep's launch duration; and what separates the two flavours'
    # results after identical iteration counts from identical starts: pose RMSE, and the association decisions of 2 x 10^5 sampled
    # (surfel, keyframe) pairs evaluated by both flavours on one state.
    other_flavour = None
    if not args.no_extras and not args.pcg and not args.intrinsics and shard_world == 1 and world == 1:
        other = "exact" if args.arithmetic == "fast" else "fast"
        capi.check(ctx.lib.bahip_set_profiling(ctx.handle, 0))
        ba.SetFastArithmetic(other == "fast")
        reset_to_start_state()
        if args.warmup > 0:
            run(args.warmup)
        ctx.synchronize()
        torch.cuda.synchronize()
        t_o = time.perf_counter()
        run(args.steps)
        ctx.synchronize()
        dt_o = time.perf_counter() - t_o
        other_stats = ba.last_stats()
        other_poses = np.array([ba.keyframe_pose(k) for k in range(args.keyframes)])
        # association decisions pair by pair, both flavours on this state
        ba.BindScene()
        surfels_struct = ba.surfels_struct()
        from badslam_amd import se3
        pair_kfs = sorted(np.random.Generator(np.random.PCG64(5)).choice(K, min(5, K), replace=False).tolist())
        decisions = {}
        for flavour in ("exact", "fast"):
            ba.SetFastArithmetic(flavour == "fast")
            pair_rng = np.random.Generator(np.random.PCG64(6))
            outs = []
            for k in pair_kfs:
                idx = pair_rng.integers(0, N_total, 40000).astype(np.uint32)
                F = np.ascontiguousarray(se3.matrix(se3.inverse(ba.keyframe_pose(k)))[:3, :4], np.float32).reshape(-1)
                pairs_out = np.zeros((len(idx), 40), np.float32)
                frame = ba.keyframe_frame(k)
                capi.check(ctx.lib.bahip_debug_evaluate_pairs(ctx.handle, C.byref(frame), F.ctypes.data_as(C.POINTER(C.c_float)), C.byref(surfels_struct),
                                                              idx.ctypes.data_as(C.POINTER(C.c_uint32)), len(idx), pairs_out.ctypes.data_as(C.POINTER(C.c_float))))
                outs.append(pairs_out[:, 0].copy())
            decisions[flavour] = np.concatenate(outs)
        associated = int(np.count_nonzero(decisions["exact"] == 1.0))
        flips = int(np.count_nonzero(decisions["exact"] != decisions["fast"]))
        # the instrumented repeat of the other flavour's timed region
        ba.SetFastArithmetic(other == "fast")
        reset_to_start_state()
        if args.warmup > 0:
            run(args.warmup)
        capi.check(ctx.lib.bahip_set_profiling(ctx.handle, 3))
        ctx.synchronize()
        run(args.steps)
        ctx.synchronize()
        o_ms, o_launches = read_stage_timers()
        o_units = C.c_longlong()
        capi.check(ctx.lib.bahip_stage_work_units(ctx.handle, 2, C.byref(o_units)))
        capi.check(ctx.lib.bahip_set_profiling(ctx.handle, 0))
        o_n = max(1, int(o_launches[2]))
        o_bytes = N_total * 28 + (o_units.value / o_n) * args.width * args.height * 5
        o_avg_ms = o_ms[2] / o_n
        dpose = np.linalg.norm(other_poses[:, 4:] - timed_end_poses[:, 4:], axis=1)
        other_flavour = {"arithmetic": other, "ba_iterations_per_s": args.steps / dt_o, "ms_per_step": 1e3 * dt_o / args.steps,
                         "frac": o_bytes / (o_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "pose_launch_ms": o_avg_ms,
                         "pose_gn_rounds_per_iteration": other_stats["pose_rounds"] / args.steps,
                         "parity": {"against": args.arithmetic + " arithmetic, same start state, same warm-up and iteration count",
                                    "pose_rmse_m": float(np.sqrt(np.mean(dpose ** 2))), "pose_max_m": float(dpose.max()),
                                    "sampled_pairs": int(len(decisions["exact"])), "associated_pairs": associated, "flips": flips,
                                    "flips_over_associated": flips / max(1, associated)},
                         "what": ("v_rcp_f32 / v_sqrt_f32 / v_exp_f32, contraction, binary32 denormals flushed in the sweeps -- the arithmetic of the reference's own "
                                  "build (nvcc -use_fast_math); sums keep their defined order" if other == "fast" else
                                  "correctly rounded reciprocal / square root / division, defined exp, no contraction beyond the spelled fused multiply-adds: "
                                  "every bit is the CPU oracle's"),
                         "note": "the other flavour of the sweeps, measured after the timed region by the same protocol (scene reset, warm-up, "
                                 f"{args.steps} timed iterations, no pre-pass of its own: the GPU is busy already); tolerance tests: tests/test_gpu_fast_flavour.py, "
                                 "tests/test_gpu_golden_reference.py[fast], tests/test_gpu_scale_parity.py::test_c3_fast_flavour_against_the_exact_build"}
        ba.SetFastArithmetic(args.arithmetic == "fast")
        reset_to_start_state()

    def intrinsics_roofline(sweep_ms):
        # the stage's dominant kernel against the HBM roof: its algorithmic bytes are the pose sweep's (surfel rows once, 5 bytes of
        # every keyframe pixel) plus the 32-byte record it writes per associated pair with a depth residual (read back by the reduction)
        sweep_bytes = N_total * 28 + K * args.width * args.height * 5
        intr_pmc, intr_source = committed_profile(args, intrinsics=True, pcg=False)
        intr_traffic = pmc_kernel_entry(intr_pmc, intr_source, "intrinsics_accumulate_kernel")
        reduce_traffic = pmc_kernel_entry(intr_pmc, intr_source, "intrinsics_bin_reduce")
        return {"bound": "hbm", "kernel": "intrinsics_accumulate_kernel<true,true>", "avg_launch_ms": sweep_ms,
                "algorithmic_bytes_per_launch": sweep_bytes, "achieved": sweep_bytes / (sweep_ms * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": sweep_bytes / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "traffic": intr_traffic["bytes"] if intr_traffic else None,
                "traffic_source": intr_traffic["source"] if intr_traffic else None,
                "record_reduction_traffic": reduce_traffic["bytes"] if reduce_traffic else None,
                "limiter": "instruction issue (1.3 x the pose sweep's VALU work; 24 of its 34 per-lane sums live in LDS: 127 VGPRs, 4 wavefronts per SIMD) and the "
                           "32-byte record it writes per associated pair with a depth residual (read back by the second "
                           "kernel, which sorts the records of a chunk by cell in LDS and adds them from registers)"}

    # Untimed extras, so that the driver's default run also sees the other two stages of SURVEY 8d: the intrinsics step of the
    # alternating scheme (reference timing key BA_intrinsics_optimization, B/direct_ba_alternating.cc:687) and the PCG scheme.
    extras = {}
    if not args.no_extras and not args.pcg and shard_world == 1 and world == 1:
        EXTRA_STEPS = 3
        if not args.intrinsics:
            capi.check(ctx.lib.bahip_set_profiling(ctx.handle, 2))
            run(EXTRA_STEPS, intrinsics=True)
            ctx.synchronize()
            ms, _ = read_stage_timers()
            pairs = None
            extras["intrinsics"] = {"BA_intrinsics_optimization_ms_per_iteration": ms[4] / EXTRA_STEPS, "iterations": EXTRA_STEPS,
                                    "sweep_ms": ms[6] / EXTRA_STEPS, "record_reduction_ms": ms[7] / EXTRA_STEPS,
                                    "sweep_ms_note": ("sweep in slices with the record reductions on a second stream behind it: sweep_ms covers both"
                                                      if ms[7] == 0 else "sweep, then the reduction of the records"),
                                    "note": "alternating iterations with depth + colour intrinsics optimisation after the timed region"}
            extras["roofline_intrinsics"] = intrinsics_roofline(ms[6] / EXTRA_STEPS)
            cc, dc, _a = ba.cameras()
            ba.set_cameras(cc, dc, 0.0)        # back to a = 0 for what follows (the cfactor image keeps its update)
        capi.check(ctx.lib.bahip_set_profiling(ctx.handle, 0))
        run(1, intrinsics=False, pcg=True)     # warm: allocates the PCG vectors
        ctx.synchronize()
        t_pcg = time.perf_counter()
        run(EXTRA_STEPS, intrinsics=False, pcg=True)
        ctx.synchronize()
        dt_pcg = time.perf_counter() - t_pcg
        inner = ba.last_stats()["pcg_inner_steps"] / EXTRA_STEPS
        # one more outer iteration with event pairs around the step-1 sweeps (the scheme's dominant kernel), outside the timed ones
        capi.check(ctx.lib.bahip_set_profiling(ctx.handle, 2))
        run(1, intrinsics=False, pcg=True)
        ctx.synchronize()
        ms_pcg, n_pcg = read_stage_timers()
        capi.check(ctx.lib.bahip_set_profiling(ctx.handle, 0))
        if n_pcg[5] > 0:
            U = 6 * (K - 1) + 3 * N_total
            step1_ms = ms_pcg[5] / n_pcg[5]
            # algorithmic bytes of one step-1 sweep: surfel rows once, p and g of the surfel block (3 + 3 floats
            # per surfel), 5 bytes of every keyframe pixel
            step1_bytes = N_total * (28 + 6 * 4) + K * args.width * args.height * 5
            pcg_pmc, pcg_source = committed_profile(args, intrinsics=False, pcg=True)
            # which form of the sweep ran (the LDS form needs >= 8192 tiles, a pose block that fits its table and the opt-in)
            tile_form, lds_form_n = C.c_longlong(), C.c_longlong()
            capi.check(ctx.lib.bahip_debug_pcg_step1_form_launches(C.byref(tile_form), C.byref(lds_form_n)))
            pcg_lds = lds_form_n.value >= tile_form.value
            pcg_traffic = pmc_kernel_entry(pcg_pmc, pcg_source, "pcg_step1_lds_kernel" if pcg_lds else "pcg_step1_kernel")
            extras["roofline_pcg"] = {"bound": "hbm", "kernel": ("pcg_step1_lds_kernel<false,false> (persistent, pose block of the dense head in LDS)" if pcg_lds
                                                                 else "pcg_step1_kernel<false,false> (one tile per wavefront, global atomics on the exact accumulators)"),
                                      "launches_by_form": {"lds": int(lds_form_n.value), "tile_per_wavefront": int(tile_form.value)},
                                      "avg_launch_ms": step1_ms, "launches": int(n_pcg[5]), "algorithmic_bytes_per_launch": step1_bytes,
                                      "achieved": step1_bytes / (step1_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": step1_bytes / (step1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                      "traffic": pcg_traffic["bytes"] if pcg_traffic else None, "traffic_source": pcg_traffic["source"] if pcg_traffic else None,
                                      "unknowns": int(U),
                                      "limiter": "instruction issue, as the pose sweep: the same pair work plus s = J p and g += J^T (w s)"}
        extras["pcg"] = {"outer_iterations_per_s": EXTRA_STEPS / dt_pcg, "ms_per_outer_iteration": 1e3 * dt_pcg / EXTRA_STEPS,
                         "inner_steps_per_outer_iteration": inner, "inner_steps_per_s": inner * EXTRA_STEPS / dt_pcg,
                         "max_inner_iterations": 30, "iterations": EXTRA_STEPS,
                         "note": "PCG scheme (poses + geometry) on the same scene after the timed region"}

    if args.intrinsics and not args.pcg and shard_world == 1 and world == 1 and breakdown_ms[6] > 0:
        # (a run WITH the intrinsics step -- BASELINE configs[4] on one GPU: the stage is half the iteration)
        extras["roofline_intrinsics"] = intrinsics_roofline(breakdown_ms[6] / BREAKDOWN_STEPS)
        extras["roofline_intrinsics"]["sweep_ms_note"] = ("stage timer 6 over the breakdown iterations; with the sweep in slices it covers the record "
                                                          "reductions that run on the second stream behind each slice")
    if not args.no_extras and not args.pcg and not args.intrinsics and shard_world == 1 and world == 1:
        # Cold start, last of the extras: `value` is measured after the warm-up iterations have absorbed the 5 mm / 1 mrad
        # perturbation (R close to 1 Gauss-Newton round per keyframe).  Here the scene is put back to its perturbed state --
        # surfels re-uploaded, poses, cameras and cfactor image reset -- and the FIRST iterations are timed: every keyframe takes several rounds.
        COLD_STEPS = 5
        capi.check(ctx.lib.bahip_set_profiling(ctx.handle, 0))
        ba.upload_surfels(data)
        for k, T in enumerate(start_poses):
            ba.set_keyframe_pose(k, T)
        ba.set_cameras(*start_cameras)             # the intrinsics extra moved the cameras and filled the cfactor image
        ba.L.dba_clear_cfactor(ba.h, ba.stream)
        ctx.synchronize()
        torch.cuda.synchronize()
        t_cold = time.perf_counter()
        run(COLD_STEPS)
        ctx.synchronize()
        dt_cold = time.perf_counter() - t_cold
        cold_stats = ba.last_stats()
        extras["cold_start"] = {"iterations": COLD_STEPS, "ba_iterations_per_s": COLD_STEPS / dt_cold, "ms_per_iteration": 1e3 * dt_cold / COLD_STEPS,
                                "pose_gn_rounds_per_iteration": cold_stats["pose_rounds"] / COLD_STEPS,
                                "pose_gn_steps_per_keyframe": cold_stats["pose_steps"] / (COLD_STEPS * K),
                                "note": "iterations 1-5 from the perturbed state (poses * exp(N(0, 5 mm / 1 mrad)), surfels + U(0, 5 mm)), after the "
                                        "timed region; `value` is the rate once the perturbation has been absorbed"}

    if not args.no_extras and not args.pcg and not args.intrinsics and shard_world == 1 and world == 1:
        capi.check(ctx.lib.bahip_set_profiling(ctx.handle, 0))

        def reset_scene(surfels):
            ba.upload_surfels(surfels)
            for k, T in enumerate(start_poses):
                ba.set_keyframe_pose(k, T)
            ba.set_cameras(*start_cameras)
            ba.L.dba_clear_cfactor(ba.h, ba.stream)

        # (a) the same protocol on the cloud in CREATION order (what round 3's callers of the reference API ran on: VERDICT r3 weak 3)
        creation_order = getattr(build_scene, "creation_order", None)
        if creation_order is not None:
            reset_scene(creation_order)
            run(args.warmup)
            ctx.synchronize()
            t_u = time.perf_counter()
            run(args.steps)
            ctx.synchronize()
            dt_u = time.perf_counter() - t_u
            extras["unsorted_ba_iterations_per_s"] = args.steps / dt_u
        # (b) the drop-in call path: nothing but methods of B/direct_ba.h:73-388 -- BundleAdjustment with the reference's defaults
        # do_surfel_updates = true and increase_ba_iteration_count = true (B/bad_slam_config.h:219, B/bad_slam.cc:261-274), ten
        # iterations per call (max_num_ba_iterations_per_keyframe), the full window of the timed region.  Every call then runs
        # the surfel lifecycle for every keyframe (filtered creation + merging at its first iteration, merging + deletion +
        # compaction + the Morton reorder in its end tasks); the cloud starts in creation order.
        if creation_order is not None:
            reset_scene(creation_order)
            DROP_CALLS, DROP_ITERATIONS = 3, 10

            def drop_in_call():
                done, _ = ba.BundleAdjustment(do_surfel_updates=True, optimize_poses=True, optimize_geometry=True, min_iterations=DROP_ITERATIONS,
                                              max_iterations=DROP_ITERATIONS, active_keyframe_window_start=0, active_keyframe_window_end=K - 1,
                                              increase_ba_iteration_count=True)
                return done

            drop_in_call()                     # the first call absorbs the perturbation and leaves the buffer in Morton order
            ctx.synchronize()
            n_before = ba.surfels_size()
            t_d = time.perf_counter()
            iterations = sum(drop_in_call() for _ in range(DROP_CALLS))
            ctx.synchronize()
            dt_d = time.perf_counter() - t_d
            # the same calls without the lifecycle and the end tasks, for the split
            ba.set_ba_iteration_counts(1, 1)
            run(DROP_ITERATIONS)
            ctx.synchronize()
            t_p = time.perf_counter()
            for _ in range(DROP_CALLS):
                run(DROP_ITERATIONS)
            ctx.synchronize()
            dt_p = time.perf_counter() - t_p
            extras["drop_in"] = {"calls": DROP_CALLS, "iterations_per_call": iterations / DROP_CALLS, "ms_per_call": 1e3 * dt_d / DROP_CALLS,
                                 "ba_iterations_per_s": iterations / dt_d,
                                 "ms_per_call_iterations_only": 1e3 * dt_p / DROP_CALLS,
                                 "lifecycle_and_end_tasks_ms_per_call": 1e3 * (dt_d - dt_p) / DROP_CALLS,
                                 "surfels": [int(n_before), int(ba.surfels_size())],
                                 "what": "vis::DirectBA::BundleAdjustment(do_surfel_updates = true, increase_ba_iteration_count = true, "
                                         f"{DROP_ITERATIONS} iterations, window 0 .. K-1) per call, cloud initially in creation order: filtered "
                                         "creation + merging for all keyframes inside the call, merging + deletion + compaction + Morton reorder "
                                         "in its end tasks (DirectBA::PerformBASchemeEndTasks); only methods of B/direct_ba.h:73-388"}
        reset_scene(data)

    per_rank = None
    if dist is not None and world > 1:
        # what makes a measured scaling curve readable: every rank's own time and stage breakdown, and what was exchanged
        mine_report = {"rank": rank, "surfels": int(mine.size), "ms_per_step_before_barrier": 1e3 * local_elapsed / args.steps,
                       "stage_ms_per_iteration": {STAGES[s]: float(breakdown_ms[s]) / BREAKDOWN_STEPS for s in range(5 if args.intrinsics else 4)},
                       "pose_accumulate_ms_per_launch_timed_region": float(stage_ms[2]) / max(1, int(stage_launches[2]))}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine_report)
    if rank == 0:
        W, H = args.width, args.height
        N_rank = int(mine.size)
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
            "config": {"workload": f"synthetic {W}x{H}, {K} keyframes, {N_total} surfels, geometry+photometric "
                                   f"{'PCG' if args.pcg else 'alternating'} BA"
                                   f"{' with intrinsics' if args.intrinsics else ''}"
                                   f"{' (BASELINE configs[2])' if (W, H, K, args.surfels, args.intrinsics) == (640, 480, 200, 3000000, False) else ''}",
                       "keyframes": K, "surfels": int(N_total), "width": W, "height": H,
                       "host": "C++ vis::DirectBA::BundleAdjustment over the bahip_* C ABI",
                       "arithmetic": (args.arithmetic + (" (bit-identical to the CPU oracle)" if args.arithmetic == "exact" else
                                                         " (v_rcp / v_sqrt / v_exp, contraction, FTZ: as the reference's -use_fast_math build; held to the reference's kernels by tolerance)")),
                       "tile_schedule": ("buffer order (BAHIP_TILE_ORDER=0)" if os.environ.get("BAHIP_TILE_ORDER") == "0"
                                         else "heavy work first, from the pose sweep's per-tile candidate counts (wave_cull.h: scheduled_tile)"),
                       "surfel_order": "creation order" if args.no_spatial_sort else "DirectBA::SortSurfelsSpatially (Morton, %g cm grid)" % (100 * args.sort_cell),
                       "parallelism": (f"keyframe-shard x{world}, RCCL all-reduce of the geometry step's class partials and of pose H,b" if by_keyframes
                                       else f"surfel-shard x{world}, RCCL all-reduce of pose H,b") if world > 1 else "single GPU"},
            **({"emulated_share_of_world": shard_world,
                "emulated_share_with_link": {
                    "exchanges_per_iteration": stats["pose_rounds"] / args.steps,
                    "assumed_us_per_exchange": [20, 40],
                    "ms_per_step_with_link": [1e3 * elapsed / args.steps + 1e-3 * us * stats["pose_rounds"] / args.steps for us in (20, 40)],
                    "note": "one rank's share of the surfels on one GPU, every exchange a no-op: no link time is in ms_per_step.  A real run adds one "
                            "all-reduce of K x 56 int64 (21.6 KB at 200 keyframes: latency-bound, SURVEY 8e: 20-40 us over xGMI) per Gauss-Newton "
                            "round; ms_per_step_with_link adds that budget.  Not a measurement of a multi-GPU run."}} if shard_world != world else {}),
            **({"exchange": {"calls_per_iteration": exchange_calls.value / args.steps, "bytes_per_iteration": exchange_bytes.value / args.steps,
                             "what": "int64 fixed-point pose normal equations, one all-reduce per Gauss-Newton round"
                                     + ("; binary64 intrinsics accumulators" if args.intrinsics else "")
                                     + ("; int64 limbs of the PCG scheme's exact sums, two per inner step" if args.pcg else ""),
                             "transport": "torch.distributed hook" if hook_keepalive is not None else "native RCCL (ncclAllReduce on the backend's stream)",
                             "torch_backend": dist.get_backend() if dist is not None else None,
                             "n_ranks_seen": ranks_seen,
                             "n_ranks_seen_note": "sum of 1 over the ranks through that transport before the first iteration (bahip_context_count_ranks, with a time limit)"},
                "per_rank": per_rank} if per_rank is not None else {}),
            "prepass": {"iterations": (args.warmup + args.steps) if prepass else 0,
                        "first_pass_ba_iterations_per_s": first_pass_rate,
                        "first_pass_note": "the pre-pass timed by the contract's protocol (W untimed iterations, then K timed): what `value` reads "
                                           "without a pre-pass, i.e. as the first sustained load of the process (this rank's clock, no barrier)",
                        "note": "the iterations of warm-up + timed region run once before and are discarded, the scene is reset to the same start "
                                "state: the timed region is then not the first sustained load of the process (clocks, first touch: +1.5-2 %); "
                                "--no-prepass switches it off"},
            "loop": {"timed_calls_driven_by_the_device": int(loop_handled1.value - loop_handled0.value),
                     "timed_calls_driven_by_the_host": int(loop_declined1.value - loop_declined0.value),
                     "note": "bahip_alternating_iterations: all iterations of a BundleAdjustment call queued at once, the stopping rule on the device, "
                             "one host wait -- or declined (surfel updates, intrinsics, PCG, keyframe sharding, a host all-reduce hook) and driven "
                             "by the host class round by round; the plain alternating bench must read 1 / 0"},
            "stage_ms_per_iteration": {STAGES[s]: breakdown_ms[s] / BREAKDOWN_STEPS for s in range(5 if args.intrinsics else 4)},
            "stage_ms_note": f"{BREAKDOWN_STEPS} further iterations after the timed region, all stages timed; the surfel activation "
                             "is decided inside the normals pass of the geometry sweep (one launch), hence 0",
            **extras,
            **({"stale_profiles_not_quoted": sorted(set(STALE_PROFILES)),
                "stale_profiles_note": "counter summaries of this workload under profiles/ whose kernel-source digest (badslam_amd/buildinfo.py) is not that of "
                                       "the sources this run was built from: their traffic figures are NOT quoted (scripts/profile_all.sh refreshes them)"}
               if STALE_PROFILES else {}),
            **({("fast_math" if other_flavour["arithmetic"] == "fast" else "exact_arithmetic"): other_flavour} if other_flavour else {}),
        }
        if not args.pcg:
            R = stats["pose_rounds"] / args.steps
            Rbar = stats["pose_steps"] / (args.steps * K)
            launches = max(1, int(stage_launches[2]))
            # algorithmic bytes of one pose-accumulate launch (DESIGN.md "pose_accumulate"): this rank's surfel
            # positions / normals / descriptors once (28 B) + the depth, normal and luma texels (5 B/pixel) of the
            # keyframes still iterating in that Gauss-Newton round, averaged over the timed launches that did work
            kf_per_launch = units.value / launches
            bytes_pose_launch = N_rank * 28 + kf_per_launch * W * H * 5
            avg_ms = stage_ms[2] / launches
            achieved = bytes_pose_launch / (avg_ms * 1e-3) / 1e9
            # One convention for the whole line: the passes as they are LAUNCHED.  The geometry launch fuses the
            # reference's activation and normals passes (one read of the depth / normal plane, 4 B per pixel, and of the surfel
            # rows, 21 B incl. the flag and the normal written back) with the position / descriptor pass (4 + 1 B per pixel,
            # 49 B per surfel); a pose launch reads 28 B per surfel and 5 B per pixel of the keyframes still iterating.
            # SURVEY 8d's formula charges the activation as a pass of its own (17 B per surfel, 4 B per pixel more): kept beside it.
            geo_bytes_surfel, geo_bytes_pixel = 21 + 49, 4 + 5
            pose_bytes_iter = launches * N_rank * 28 / args.steps + (units.value / args.steps) * W * H * 5
            b_alg_iter = N_total * geo_bytes_surfel + K * W * H * geo_bytes_pixel + (pose_bytes_iter if N_rank == N_total else N_total * 28 * R + K * W * H * 5 * Rbar)
            b_alg_survey = N_total * (17 + 21 + 49 + 28 * R) + K * W * H * (4 + 4 + 5 + 5 * Rbar)
            out["config"].update({"pose_gn_rounds_per_iteration": R, "pose_gn_steps_per_keyframe": Rbar})
            out["algorithmic_bytes_per_iteration"] = b_alg_iter
            out["algorithmic_bytes_per_iteration_by_survey_8d_formula"] = b_alg_survey
            out["iteration_fraction_of_hbm_roofline"] = b_alg_iter / (elapsed / args.steps) / (HBM_PEAK_GBS * 1e9)
            out["launch_window"] = {"pose_dispatches_before": dispatches_before, "pose_dispatches_timed": dispatches_timed,
                                    "pose_launches_with_work_timed": launches, "keyframes_visited_timed": int(units.value),
                                    "geometry_dispatches_before": 2 * args.warmup + args.steps + ((args.warmup + args.steps) if prepass else 0), "geometry_dispatches_timed": args.steps,
                                    "iterations_timed": args.steps, "surfels": N_rank,
                                    "note": "which dispatches of the sweeps belong to the instrumented repeat of the timed region (in process "
                                            "order; the timed region's own iterations count as `before`); "
                                            "scripts/summarize_profile.py sums rocprofv3's per-dispatch rows over exactly these"}
            out["instrumented_region"] = {"ms_per_step": 1e3 * instrumented_elapsed / args.steps,
                                          "ba_iterations_per_s": args.steps / instrumented_elapsed,
                                          "slowdown_by_event_records": instrumented_elapsed / local_elapsed - 1.0,
                                          "same_work_as_timed_region": (repeat_stats["pose_rounds"], repeat_stats["pose_steps"]) == (stats["pose_rounds"], stats["pose_steps"]),
                                          "note": "the timed region's iterations repeated from the same start state (scene reset, same warm-up: "
                                                  "the same Gauss-Newton rounds and steps) with a hipEvent pair around every pose-accumulate launch: "
                                                  "where roofline.avg_launch_ms is measured; `value` has no event record in it"}
            pmc, pmc_source = committed_profile(args) if (world == 1 and shard_world == 1) else (None, None)
            # the pose sums have two kernels (kernels_pose.hip): persistent workgroups with the normal equations in LDS when the
            # table of the launch's work items fits, one tile per wavefront with global atomics otherwise; the line names the one
            # most of the timed launches used
            lds = form_lds.value >= form_global.value
            # (the sweeps exist once per arithmetic flavour, in namespace bahip::exact / bahip::fast: the name says which one ran)
            pose_kernel = args.arithmetic + "::" + ("pose_accumulate_lds_kernel<true, true, false>" if lds else "pose_accumulate_kernel<true, true>")
            # Counter evidence: sums over the dispatches of the PROFILE RUN's own timed region, divided by that run's own
            # launches / keyframes visited / iterations (profiles/*_pmc_per_kernel.json "timed_window") -- like by like; the
            # per-launch averages over all 40-odd dispatches of a profile run (warm-up rounds, rounds queued in vain) that
            # round 4's line divided by the timed region's figures are not used any more.
            win = (pmc or {}).get("timed_window", {}).get("pose") if pmc else None
            cal = (pmc or {}).get("fetch_calibration") if pmc else None
            pose_counters = None
            if win and cal and win.get("launches_with_work"):
                traffic_total = win["FETCH_SIZE_kb"] * 1024.0 * cal["factor"] + win.get("WRITE_SIZE_kb", 0.0) * 1024.0
                pose_counters = {
                    "source": pmc_source, "profile_iterations": win["iterations"], "profile_dispatches": win["dispatches"],
                    "profile_launches_with_work": win["launches_with_work"],
                    "traffic_per_launch": traffic_total / win["launches_with_work"],
                    "traffic_per_iteration": traffic_total / win["iterations"],
                    "algorithmic_bytes_per_iteration": win["algorithmic_bytes"] / win["iterations"],
                    "traffic_over_algorithmic": traffic_total / win["algorithmic_bytes"],
                    "fetch_factor": cal["factor"]}
                if win.get("duration_ns") and win.get("fp32_flops"):
                    pose_counters["fp32_tflops"] = win["fp32_flops"] / (win["duration_ns"] * 1e-9) / 1e12
                    pose_counters["frac_of_vector_peak"] = pose_counters["fp32_tflops"] / FP32_VECTOR_PEAK_TFLOPS
                for key in ("non_arithmetic_valu_fraction", "valu_cycles_per_instruction", "valu_instructions_per_keyframe_visit",
                            "shader_clock_mhz", "valu_fraction_int32", "valu_fraction_int64", "valu_fraction_cvt"):
                    if key in win:
                        pose_counters[key] = win[key]
            out["roofline"] = {"bound": "hbm", "kernel": pose_kernel.replace(", ", ","), "achieved": achieved,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                               "traffic": pose_counters["traffic_per_launch"] if pose_counters else None,
                               "traffic_source": pose_counters["source"] if pose_counters else None,
                               "traffic_fetch_factor": pose_counters["fetch_factor"] if pose_counters else None,
                               "traffic_over_algorithmic": pose_counters["traffic_over_algorithmic"] if pose_counters else None,
                               "algorithmic_bytes_per_launch": bytes_pose_launch, "avg_launch_ms": avg_ms, "launches": launches,
                               "launches_by_form": {"lds": int(form_lds.value), "global_atomics": int(form_global.value)},
                               "keyframes_per_launch": kf_per_launch,
                               "limiter": "instruction issue, not HBM: the sweep is several hundred VALU + ~150 scalar instructions per visited "
                                          "(surfel tile, keyframe) candidate at 4 wavefronts per SIMD; a wave64 VALU instruction occupies "
                                          "the SIMD-32 for 2 cycles at best (scripts/microbench/inst_cost.hip), DESIGN.md section 5",
                               "counters": pose_counters,
                               # binary32 flops of the VALU instruction mix (PMC: 64 lanes x (add + mul + transcendental + 2 fma) wave
                               # instructions, masked lanes included) of the profile run's timed region over the kernel time of the
                               # same dispatches in that run, against the 157.3 TFLOP/s vector peak; and the share of VALU
                               # instructions that are not floating-point arithmetic
                               "fp32_tflops": pose_counters.get("fp32_tflops") if pose_counters else None,
                               "frac_of_vector_peak": pose_counters.get("frac_of_vector_peak") if pose_counters else None,
                               "non_arithmetic_valu_fraction": pose_counters.get("non_arithmetic_valu_fraction") if pose_counters else None,
                               "valu_cycles_per_instruction": pose_counters.get("valu_cycles_per_instruction") if pose_counters else None}
            # the other sweep of an iteration: activation + normals + position/descriptor step in one launch
            geo_ms = breakdown_ms[1] / BREAKDOWN_STEPS
            bytes_geo = N_rank * geo_bytes_surfel + K * W * H * geo_bytes_pixel
            gwin = (pmc or {}).get("timed_window", {}).get("geometry") if pmc else None
            geo_counters = None
            if gwin and cal and gwin.get("dispatches"):
                g_total = gwin["FETCH_SIZE_kb"] * 1024.0 * cal["factor"] + gwin.get("WRITE_SIZE_kb", 0.0) * 1024.0
                geo_counters = {"source": pmc_source, "traffic_per_launch": g_total / gwin["dispatches"],
                                "traffic_over_algorithmic": g_total / gwin["dispatches"] / bytes_geo}
                if gwin.get("duration_ns") and gwin.get("fp32_flops"):
                    geo_counters["fp32_tflops"] = gwin["fp32_flops"] / (gwin["duration_ns"] * 1e-9) / 1e12
                for key in ("non_arithmetic_valu_fraction", "valu_cycles_per_instruction", "shader_clock_mhz"):
                    if key in gwin:
                        geo_counters[key] = gwin[key]
            out["roofline_geometry"] = {"bound": "hbm", "kernel": args.arithmetic + "::geometry_kernel<true,true> (activation + normals + position step)",
                                        "achieved": bytes_geo / (geo_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": bytes_geo / (geo_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        "traffic": geo_counters["traffic_per_launch"] if geo_counters else None,
                                        "traffic_over_algorithmic": geo_counters["traffic_over_algorithmic"] if geo_counters else None,
                                        "algorithmic_bytes_per_launch": bytes_geo, "avg_launch_ms": geo_ms,
                                        "counters": geo_counters,
                                        "fp32_tflops": geo_counters.get("fp32_tflops") if geo_counters else None,
                                        "non_arithmetic_valu_fraction": geo_counters.get("non_arithmetic_valu_fraction") if geo_counters else None}
        else:
            out["config"]["pcg_inner_steps_per_iteration"] = stats["pcg_inner_steps"] / args.steps
        if not args.no_cpu_baseline and world == 1:
            cb = cpu_baseline(args, ba, data, log)
            sweeps = 3 + (stats["pose_rounds"] / args.steps if not args.pcg else 3)
            out["cpu_baseline"] = {"value": cb["pairs_per_s"], "unit": "surfel-keyframe pairs/s (full cost evaluation)",
                                   "cores": cb["cores"], "kind": cb["kind"],
                                   "sample": f"ONE full cost evaluation of the bench scene itself by {cb['who']}: {cb['K']} keyframes x "
                                             f"{cb['N']} surfels {W}x{H} = {cb['K'] * cb['N']:.3g} pairs, {cb['nres']} residuals, "
                                             f"{cb['seconds_per_eval']:.2f} s; nothing extrapolated",
                                   "seconds_per_cost_evaluation": cb["seconds_per_eval"],
                                   **({"port": {"value": cb["port"]["pairs_per_s"], "unit": "surfel-keyframe pairs/s (full cost evaluation)",
                                                "cores": cb["port"]["cores"], "seconds_per_cost_evaluation": cb["port"]["seconds_per_eval"],
                                                "residuals": cb["port"]["nres"],
                                                "what": "the same evaluation by " + cb["port"]["who"]}} if "port" in cb else {}),
                                   "ba_iteration_lower_bound_note": f"one BA iteration makes >= {sweeps:.1f} such sweeps (activation, normals, "
                                                                    "position step, pose rounds) plus the Jacobians",
                                   "equivalent_ba_iterations_per_s": 1.0 / (sweeps * cb["seconds_per_eval"])}
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
