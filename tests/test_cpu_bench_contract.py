"""The committed bench line of the latest profile (profiles/<round>_bench.json) against the driver's contract, and against
the rocprofv3 summary of the same command committed beside it: the live hipEvent duration of the dominant kernel must agree
with the trace's average, `achieved` must be the algorithmic bytes over that duration, `frac` the ratio to the 8 TB/s peak."""
import csv
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest(pattern):
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", pattern)) if "config" not in f and "pcg" not in f and "intr" not in f)
    assert files, pattern
    return files[-1]


@pytest.fixture(scope="module")
def line():
    return json.load(open(_latest("*_bench.json")))


def test_contract_fields(line):
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert line["metric"].replace("x", "×") == base["metric"].replace("x", "×")
    assert line["unit"] == "BA iterations/s" and line["higher_is_better"] is True
    assert line["n_gpus"] == 1 and line["steps"] == 20 and line["warmup"] == 3
    assert abs(line["value"] * line["ms_per_step"] - 1e3) < 1e-6 * 1e3                 # value = iterations / s of the timed region
    assert line["scaling"] in ("weak", "strong") and line["vs_baseline"] is None and line["data"] == "synthetic"
    assert line["dtype"] == "f32"
    cfg = line["config"]
    assert "workload" in cfg and "200 keyframes" in cfg["workload"] and "3000000 surfels" in cfg["workload"] and "640x480" in cfg["workload"]
    assert not any(k in cfg for k in ("model", "global_batch", "seq_len"))
    assert line["value"] >= 30                                                          # the north-star target on one GPU
    # (from r5_e on) the timed call was driven by the device: DirectBA::BundleAdjustment -> bahip_alternating_iterations handled it
    if "loop" in line:
        assert line["loop"]["timed_calls_driven_by_the_device"] == 1 and line["loop"]["timed_calls_driven_by_the_host"] == 0


def test_roofline_object_is_consistent(line):
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] > 0.5 * r["algorithmic_bytes_per_launch"]
    # the rocprofv3 --kernel-trace --stats summary of the same command, committed beside the bench line
    stats = _latest("*_kernel_stats.csv")
    assert os.path.basename(stats).split("_kernel")[0] == os.path.basename(_latest("*_bench.json")).split("_bench")[0]
    rows = [row for row in csv.DictReader(open(stats)) if row["Name"].replace(" ", "").startswith("voidbahip::" + r["kernel"].replace(" ", ""))
            or row["Name"].replace(" ", "").startswith(r["kernel"].replace(" ", ""))]
    assert len(rows) == 1, [row["Name"][:60] for row in rows]
    traced_ms = float(rows[0]["AverageNs"]) * 1e-6
    # Since round 4 the rounds of a pose phase are queued ahead of the host, and a round queued in vain is a launch of this kernel
    # that returns at once: rocprofv3 averages over EVERY launch, the roofline object over the launches that did work.  The
    # accounting of the same trace (scripts/pose_launch_accounting.py -> <tag>_pose_launches.json) connects the two.
    accounting = os.path.join(os.path.dirname(stats), os.path.basename(stats).split("_kernel")[0] + "_pose_launches.json")
    if os.path.exists(accounting):
        a = json.load(open(accounting))
        assert a["all_launches"]["n"] == int(rows[0]["Calls"]) and abs(a["all_launches"]["avg_us"] * 1e-3 - traced_ms) < 1e-3 * traced_ms
        assert a["queued_in_vain"]["n"] + a["worked"]["n"] == a["all_launches"]["n"] and a["queued_in_vain"]["avg_us"] < 10.0
        traced_ms = a["worked"]["avg_us"] * 1e-3
    assert abs(traced_ms - r["avg_launch_ms"]) < 0.05 * r["avg_launch_ms"], (traced_ms, r["avg_launch_ms"])


def test_cpu_baseline_object(line):
    c = line["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    assert isinstance(c["sample"], str) and "oracle" in c["sample"] and isinstance(c["unit"], str)


def test_bench_parses_its_arguments_without_a_gpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout


def _bench(*flags, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, timeout=300, env=e, cwd=ROOT)


def test_gpus_flag_is_honoured_or_fails_loudly():
    """`bench.py --gpus N` launches N ranks itself when no launcher did.  Without N devices it must refuse -- a run that
    silently uses one GPU and prints "n_gpus": 1 would be read as a scaling result (VERDICT r2 missing 2).  This box has no
    HIP device at all, so both the self-launch path and a mismatching launcher must end non-zero with a message and no line."""
    out = _bench("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert out.returncode != 0 and "needs 2 HIP devices" in out.stderr and out.stdout.strip() == ""
    out = _bench("--gpus", "8", "--steps", "1", "--warmup", "0", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr and out.stdout.strip() == ""
    out = _bench("--gpus", "0")
    assert out.returncode != 0
