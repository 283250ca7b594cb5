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
