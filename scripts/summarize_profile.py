#!/usr/bin/env python3
"""Condenses a scripts/profile_round.sh output directory:
  <dir>/kernel_stats.csv        copy of rocprofv3's per-kernel time table (pass 1)
  <dir>/pmc_per_kernel.json     {"config": the bench workload, "kernels": per-kernel per-launch counter averages (KB for
                                FETCH_SIZE / WRITE_SIZE as rocprofv3 reports them) + valu_issue_fraction,
                                "fetch_calibration": bytes really read per byte FETCH_SIZE reports, measured on 4-byte loads}
and prints a table.  bench.py reads the JSON (only when its config equals the run's) for roofline.traffic."""
import argparse
import collections
import csv
import glob
import json
import os
import shutil
import sys


def short(name):
    return name.split("(")[0].replace("bahip::", "").replace("(anonymous namespace)::", "").replace("void ", "")[:70]


def counters_of(d, sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for path in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                a = acc[short(r["Kernel_Name"])][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    return {k: {c: {"avg_per_launch": t / n, "launches": n} for c, (t, n) in cs.items()} for k, cs in acc.items()}


def main():
    d, tag = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "r2")
    p = argparse.ArgumentParser()
    for name, default in (("keyframes", 200), ("surfels", 3000000), ("width", 640), ("height", 480)):
        p.add_argument("--" + name, type=int, default=default)
    p.add_argument("--intrinsics", action="store_true")
    p.add_argument("--pcg", action="store_true")
    a, _ = p.parse_known_args(sys.argv[3:])
    config = {"keyframes": a.keyframes, "surfels": a.surfels, "width": a.width, "height": a.height, "intrinsics": a.intrinsics, "pcg": a.pcg}

    stats = glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(d, "kernel_stats.csv"))
        with open(stats[0]) as f:
            rows = list(csv.DictReader(f))
        print(f"{'kernel':72s} {'calls':>7s} {'avg us':>10s} {'total ms':>10s} {'%':>6s}")
        for r in rows[:16]:
            print(f"{short(r['Name']):72s} {int(r['Calls']):7d} {float(r['AverageNs']) / 1e3:10.1f} "
                  f"{float(r['TotalDurationNs']) / 1e6:10.2f} {float(r['Percentage']):6.2f}")

    kernels = {}
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_mix", "pmc_flops"):
        for k, cs in counters_of(d, sub).items():
            kernels.setdefault(k, {}).update(cs)
    for k, cs in kernels.items():
        if "SQ_INSTS_VALU" in cs and "GRBM_GUI_ACTIVE" in cs and cs["GRBM_GUI_ACTIVE"]["avg_per_launch"] > 0:
            # 256 CUs x 4 SIMDs; GRBM_GUI_ACTIVE sums the 8 XCDs.  Issue ceiling: scripts/microbench/valu_rate.hip measures
            # 2.7 cycles per wave64 binary32 VALU instruction per SIMD with 4 resident wavefronts (independent or dependent
            # v_fma_f32 alike; 2.54 with 8), i.e. the SIMD issues one every ~2.7 cycles at best -- not every 4 as r1 assumed.
            cycles = cs["GRBM_GUI_ACTIVE"]["avg_per_launch"] / 8.0
            cs["valu_cycles_per_instruction"] = 1024.0 * cycles / cs["SQ_INSTS_VALU"]["avg_per_launch"]
            cs["valu_issue_fraction"] = 2.7 / cs["valu_cycles_per_instruction"]

    for k, cs in kernels.items():
        if all(("SQ_INSTS_VALU_" + c) in cs for c in ("ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32")) and cs.get("SQ_INSTS_VALU", {}).get("avg_per_launch", 0) > 0:
            # wave64 instruction counts; an FMA is two flops, all 64 lanes counted (an upper bound: masked lanes included)
            n = {c: cs["SQ_INSTS_VALU_" + c]["avg_per_launch"] for c in ("ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32")}
            cs["fp32_flops_per_launch"] = 64.0 * (n["ADD_F32"] + n["MUL_F32"] + n["TRANS_F32"] + 2.0 * n["FMA_F32"])
            arithmetic = sum(n.values()) + sum(cs.get("SQ_INSTS_VALU_" + c, {"avg_per_launch": 0.0})["avg_per_launch"]
                                               for c in ("ADD_F64", "MUL_F64", "FMA_F64"))
            cs["non_arithmetic_valu_fraction"] = 1.0 - arithmetic / cs["SQ_INSTS_VALU"]["avg_per_launch"]
            for c in ("INT32", "INT64", "CVT"):
                if "SQ_INSTS_VALU_" + c in cs:
                    cs["valu_fraction_" + c.lower()] = cs["SQ_INSTS_VALU_" + c]["avg_per_launch"] / cs["SQ_INSTS_VALU"]["avg_per_launch"]

    cal = None
    cal_counters = counters_of(d, "pmc_cal").get("read_pattern_kernel")
    try:
        cal_bytes = int(open(os.path.join(d, "cal_bytes.txt")).read().split()[-1])
    except Exception:
        cal_bytes = 0
    if cal_counters and cal_bytes:
        # 6 launches: 3 x pattern 0 (every byte read) then 3 x pattern 1 (4 of every 128 bytes used); the average over all six
        # is reported per launch, so take the per-pattern figures from the raw rows
        rows = []
        for path in glob.glob(os.path.join(d, "pmc_cal", "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                rows += [r for r in csv.DictReader(f) if "read_pattern_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
        rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
        if len(rows) == 6:
            coalesced = sum(float(r["Counter_Value"]) for r in rows[:3]) / 3 * 1024.0
            gather = sum(float(r["Counter_Value"]) for r in rows[3:]) / 3 * 1024.0
            cal = {"factor": cal_bytes / coalesced, "buffer_bytes": cal_bytes, "coalesced_dword_reads_reported_bytes": coalesced,
                   "one_dword_per_128B_line_reported_bytes": gather,
                   "one_dword_per_128B_line_bytes_moved_at_that_factor": gather * cal_bytes / coalesced,
                   "note": "factor = bytes read / bytes FETCH_SIZE reported, for global_load_dword over a 2 GiB buffer read once"}
    out = {"config": config, "tag": tag, "kernels": kernels, "fetch_calibration": cal}
    with open(os.path.join(d, "pmc_per_kernel.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("\nfetch calibration:", json.dumps(cal))
    print("\nPMC per launch (rocprofv3 units: FETCH_SIZE / WRITE_SIZE in KB, uncorrected):")
    for k in sorted(kernels, key=lambda k: -kernels[k].get("FETCH_SIZE", {"avg_per_launch": 0})["avg_per_launch"])[:10]:
        cs = kernels[k]
        print(f"{k:60s} " + "  ".join(f"{c}={v['avg_per_launch']:.4g}" if isinstance(v, dict) else f"{c}={v:.3f}" for c, v in sorted(cs.items())))
    print(f"\ncp {d}/kernel_stats.csv profiles/{tag}_kernel_stats.csv; cp {d}/pmc_per_kernel.json profiles/{tag}_pmc_per_kernel.json; "
          f"cp {d}/summary.txt profiles/{tag}_summary.txt; cp {d}/bench_stats.json profiles/{tag}_bench_under_rocprof.json")


if __name__ == "__main__":
    main()
