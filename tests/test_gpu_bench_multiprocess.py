"""bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, one process per rank), on ONE GPU:
BENCH_DIST_BACKEND=gloo lets the ranks share the device (RCCL refuses that) and stages the all-reduce through the host.
Checks the multi-process plumbing end to end: rank / world from the environment, surfel sharding, the all-reduce hook inside
DirectBA::BundleAdjustment, barriers, max-over-ranks timing, and exactly one JSON line on stdout from rank 0."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 3])
def test_bench_under_torchrun_with_shared_device(world):
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29700 + world), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
           "--keyframes", "16", "--surfels", "150000", "--no-cpu-baseline"]
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [l for l in proc.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, proc.stdout[-2000:]                      # one JSON line, from rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["steps"] == 3 and out["scaling"] == "strong"
    assert out["value"] > 0 and out["config"]["surfels"] <= 150000
    assert "roofline" in out and out["roofline"]["launches"] >= 3   # >= one pose round per iteration on rank 0
    assert out["exchange"]["calls_per_iteration"] >= 1 and out["exchange"]["bytes_per_iteration"] >= 16 * 28 * 8
    assert [r["rank"] for r in out["per_rank"]] == list(range(world)) and sum(r["surfels"] for r in out["per_rank"]) == out["config"]["surfels"]
    # what makes the first real multi-GPU run readable: how many ranks the first exchange counted through the
    # transport the loop uses, which transport that is, and every rank's own time
    assert out["exchange"]["n_ranks_seen"] == world and out["exchange"]["torch_backend"] == "gloo"
    assert out["exchange"]["transport"] == "torch.distributed hook"          # gloo: the hook; over RCCL the native path
    assert all(r["ms_per_step_before_barrier"] > 0 and "pose_accumulate" in r["stage_ms_per_iteration"] for r in out["per_rank"])


def test_bench_sharded_by_keyframes():
    """`--shard keyframes` (BASELINE configs[3] as written) under torch.distributed.run, two ranks sharing the device over gloo:
    every rank holds the whole cloud, the geometry step's class partials and the pose normal equations are exchanged; the line
    names the axis.  Three ranks are refused (four keyframe classes)."""
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo")
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--master-addr", "127.0.0.1"]
    flags = ["--steps", "3", "--warmup", "1", "--keyframes", "16", "--surfels", "150000", "--no-cpu-baseline", "--shard", "keyframes"]
    proc = subprocess.run(base + ["--nproc-per-node=2", "--master-port", "29711", os.path.join(ROOT, "bench.py"), "--gpus", "2", *flags],
                          capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [l for l in proc.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, proc.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["config"]["parallelism"].startswith("keyframe-shard x2")
    N = out["config"]["surfels"]
    assert all(r["surfels"] == N for r in out["per_rank"])                       # the whole cloud on every rank
    # per iteration: two exchanges of class partials (4 x 5 and 4 x 8 binary32 values per surfel) + one of 16 x 56 int64 per pose round
    assert out["exchange"]["calls_per_iteration"] >= 3 and out["exchange"]["bytes_per_iteration"] >= N * 4 * 13 * 4
    proc = subprocess.run(base + ["--nproc-per-node=3", "--master-port", "29712", os.path.join(ROOT, "bench.py"), "--gpus", "3", *flags],
                          capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert proc.returncode != 0 and "2, 4 or 8 ranks" in proc.stderr


def test_bench_sharded_by_keyframes_over_eight_ranks():
    """BASELINE configs[3] as written -- "sharded by keyframe across 8 MI355X" -- as a launch: eight ranks (sharing the one device
    over gloo here), the 8-class definition of the per-surfel sums, one JSON line that says n_gpus = 8."""
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1", "--master-port", "29713",
           os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--keyframes", "16", "--surfels", "100000", "--no-cpu-baseline",
           "--shard", "keyframes"]
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [l for l in proc.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, proc.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["value"] > 0 and out["config"]["parallelism"].startswith("keyframe-shard x8")
    assert len(out["per_rank"]) == 8 and all(r["surfels"] == out["config"]["surfels"] for r in out["per_rank"])
    assert out["exchange"]["n_ranks_seen"] == 8 and out["exchange"]["calls_per_iteration"] >= 3


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (how the driver starts the N = 1 leg): bench.py starts the two
    ranks itself.  Over RCCL that needs two devices -- this box has one, so it must refuse; with BENCH_DIST_BACKEND=gloo the
    ranks share the device and the line says n_gpus = 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    flags = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--keyframes", "12", "--surfels", "100000", "--no-cpu-baseline"]
    import torch
    if torch.cuda.device_count() < 2:
        proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
        assert proc.returncode != 0 and "needs 2 HIP devices" in proc.stderr and proc.stdout.strip() == ""
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, timeout=600,
                          env=dict(env, BENCH_DIST_BACKEND="gloo"), cwd=ROOT)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [l for l in proc.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, proc.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and len(out["per_rank"]) == 2


def test_first_collective_watchdog_names_the_step():
    """A rank whose peers never arrive must end with a message, not hang: one process claims to be rank 0 of 2 and nobody else
    comes to the rendezvous."""
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29719",
               BENCH_COLLECTIVE_TIMEOUT_S="8")
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--keyframes", "8",
                           "--surfels", "50000", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert proc.returncode != 0 and proc.stdout.strip() == ""
    assert "did not finish within" in proc.stderr or "rendezvous" in proc.stderr.lower() or "timeout" in proc.stderr.lower(), proc.stderr[-2000:]
