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
    # (the sweeps live in bahip::exact / bahip::fast, one arithmetic flavour each; a profile run uses one of them: config["arithmetic"])
    return name.split("(")[0].replace("bahip::", "").replace("exact::", "").replace("fast::", "").replace("(anonymous namespace)::", "").replace("void ", "")[:70]


def counters_of(d, sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for path in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                a = acc[short(r["Kernel_Name"])][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    return {k: {c: {"avg_per_launch": t / n, "launches": n} for c, (t, n) in cs.items()} for k, cs in acc.items()}


POSE_KERNELS = ("pose_accumulate_lds_kernel", "pose_accumulate_kernel")
GEOMETRY_KERNELS = ("geometry_kernel",)


def bench_line(path):
    try:
        with open(path) as f:
            lines = [l for l in f.read().splitlines() if l.startswith("{")]
        return json.loads(lines[-1]) if lines else None
    except OSError:
        return None


def window_rows(rows, kernels, before, count):
    """The rows (one per dispatch and counter / one per dispatch of a kernel trace) of `kernels` in dispatch order, cut to the
    dispatches [before, before + count) of those kernels: the timed region of the bench run that produced them."""
    mine = [r for r in rows if short(r.get("Kernel_Name", r.get("Name", ""))).startswith(kernels)]
    ids = sorted({int(r["Dispatch_Id"]) for r in mine})
    keep = set(ids[before:before + count])
    return [r for r in mine if int(r["Dispatch_Id"]) in keep], len(keep)


def timed_window(d):
    """Counter SUMS over the dispatches of each pass's own timed region (bench.py "launch_window"), with that run's own launch /
    keyframe / iteration counts beside them -- what bench.py divides like by like (per-launch averages over all
    dispatches of a profile run mix warm-up rounds, full rounds and rounds queued in vain)."""
    out = {}
    passes = {"pmc_fetch": "bench_pmc_fetch.json", "pmc_write": "bench_pmc_write.json", "pmc_sq": "bench_pmc_sq.json",
              "pmc_flops": "bench_pmc_flops.json", "pmc_mix": "bench_pmc_mix.json"}
    per_pass = {}
    for sub, bench_file in passes.items():
        line = bench_line(os.path.join(d, bench_file))
        if not line or "launch_window" not in line:
            continue
        w = line["launch_window"]
        rows = []
        for path in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                rows += list(csv.DictReader(f))
        per_pass[sub] = w
        trace = []
        for path in glob.glob(os.path.join(d, sub, "**", "*kernel_trace.csv"), recursive=True):
            with open(path) as f:
                trace += list(csv.DictReader(f))
        for name, kernels, before, count in (("pose", POSE_KERNELS, w["pose_dispatches_before"], w["pose_dispatches_timed"]),
                                             ("geometry", GEOMETRY_KERNELS, w["geometry_dispatches_before"], w["geometry_dispatches_timed"])):
            sel, n = window_rows(rows, kernels, before, count)
            e = out.setdefault(name, {"passes": {}})
            sums = collections.defaultdict(float)
            for r in sel:
                sums[r["Counter_Name"]] += float(r["Counter_Value"])
            e["passes"][sub] = {"dispatches": n, "iterations": w["iterations_timed"], "sums": dict(sums),
                                "launches_with_work": w["pose_launches_with_work_timed"] if name == "pose" else n,
                                "keyframes_visited": w["keyframes_visited_timed"] if name == "pose" else None, "surfels": w["surfels"]}
            tsel, _ = window_rows(trace, kernels, before, count)
            if tsel:   # kernel time of the same dispatches in this (counter) pass
                e["passes"][sub]["duration_ns"] = sum(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in tsel)
    # kernel time of the same dispatches in the stats pass (its own bench line)
    stats_line = bench_line(os.path.join(d, "bench_stats.json"))
    if stats_line and "launch_window" in stats_line:
        w = stats_line["launch_window"]
        rows = []
        for path in glob.glob(os.path.join(d, "stats", "**", "*kernel_trace.csv"), recursive=True):
            with open(path) as f:
                rows += list(csv.DictReader(f))
        for name, kernels, before, count in (("pose", POSE_KERNELS, w["pose_dispatches_before"], w["pose_dispatches_timed"]),
                                             ("geometry", GEOMETRY_KERNELS, w["geometry_dispatches_before"], w["geometry_dispatches_timed"])):
            sel, n = window_rows(rows, kernels, before, count)
            if sel:
                e = out.setdefault(name, {"passes": {}})
                e["passes"]["stats"] = {"dispatches": n, "iterations": w["iterations_timed"],
                                        "duration_ns": sum(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in sel)}
    return out, stats_line


def flatten_window(name, e, config):
    """One record per sweep: every figure per ITERATION-normalised pass, scaled to the FETCH pass's iteration count."""
    p = e["passes"]
    if "pmc_fetch" not in p:
        return None
    ref = p["pmc_fetch"]
    it = ref["iterations"]

    def scaled(sub, counter):   # a pass may have run another number of iterations: per iteration, times the reference pass's count
        q = p.get(sub)
        if not q or counter not in q["sums"] or not q["iterations"]:
            return None
        return q["sums"][counter] * it / q["iterations"]

    rec = {"iterations": it, "dispatches": ref["dispatches"], "launches_with_work": ref["launches_with_work"],
           "FETCH_SIZE_kb": ref["sums"].get("FETCH_SIZE", 0.0), "WRITE_SIZE_kb": scaled("pmc_write", "WRITE_SIZE") or 0.0}
    W, H, N = config["width"], config["height"], ref["surfels"]
    if name == "pose":
        rec["keyframes_visited"] = ref["keyframes_visited"]
        rec["algorithmic_bytes"] = ref["launches_with_work"] * N * 28.0 + ref["keyframes_visited"] * W * H * 5.0
    else:
        rec["algorithmic_bytes"] = ref["dispatches"] * (N * 70.0 + config["keyframes"] * W * H * 9.0)
    if "stats" in p and p["stats"]["iterations"]:
        rec["duration_ns"] = p["stats"]["duration_ns"] * it / p["stats"]["iterations"]
    valu = scaled("pmc_flops", "SQ_INSTS_VALU") or scaled("pmc_sq", "SQ_INSTS_VALU")
    n = {c: scaled("pmc_flops", "SQ_INSTS_VALU_" + c) for c in ("ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32")}
    if valu and all(v is not None for v in n.values()):
        rec["SQ_INSTS_VALU"] = valu
        rec["fp32_flops"] = 64.0 * (n["ADD_F32"] + n["MUL_F32"] + n["TRANS_F32"] + 2.0 * n["FMA_F32"])
        f64 = sum(scaled("pmc_mix", "SQ_INSTS_VALU_" + c) or 0.0 for c in ("ADD_F64", "MUL_F64", "FMA_F64"))
        rec["non_arithmetic_valu_fraction"] = 1.0 - (sum(n.values()) + f64) / valu
        for c in ("INT32", "INT64", "CVT"):
            v = scaled("pmc_mix", "SQ_INSTS_VALU_" + c)
            if v is not None:
                rec["valu_fraction_" + c.lower()] = v / valu
        if name == "pose" and rec.get("keyframes_visited"):
            # wave-level VALU instructions per (64-surfel tile, keyframe) visit, had every tile been a candidate of every visited
            # keyframe: an upper-bound denominator; the cull leaves ~20 candidates per tile of the bench scene
            rec["valu_instructions_per_keyframe_visit"] = valu / rec["keyframes_visited"]
    cycles = scaled("pmc_sq", "GRBM_GUI_ACTIVE")
    valu_sq = scaled("pmc_sq", "SQ_INSTS_VALU")
    if cycles and valu_sq:
        # GRBM_GUI_ACTIVE sums the 8 XCDs; 256 CUs x 4 SIMDs.  A wave64 VALU instruction occupies a SIMD-32 for 2 cycles at best.
        rec["valu_cycles_per_instruction"] = 1024.0 * (cycles / 8.0) / valu_sq
        dur = p.get("pmc_sq", {}).get("duration_ns")
        if dur:   # shader cycles over the kernel time of the same dispatches in the same pass: the clock the sweep ran at
            rec["shader_clock_mhz"] = (cycles / 8.0) / (dur * it / p["pmc_sq"]["iterations"] * 1e-9) / 1e6
    return rec


def main():
    d, tag = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "r2")
    p = argparse.ArgumentParser()
    for name, default in (("keyframes", 200), ("surfels", 3000000), ("width", 640), ("height", 480)):
        p.add_argument("--" + name, type=int, default=default)
    p.add_argument("--intrinsics", action="store_true")
    p.add_argument("--pcg", action="store_true")
    p.add_argument("--arithmetic", default=None)
    a, _ = p.parse_known_args(sys.argv[3:])
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from badslam_amd import buildinfo
    arithmetic = a.arithmetic or os.environ.get("BENCH_ARITHMETIC", bench.DEFAULT_ARITHMETIC)
    config = {"keyframes": a.keyframes, "surfels": a.surfels, "width": a.width, "height": a.height, "intrinsics": a.intrinsics, "pcg": a.pcg,
              "arithmetic": arithmetic}

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
    windows, stats_line = timed_window(d)
    flat = {name: flatten_window(name, e, config) for name, e in windows.items()}
    flat = {k: v for k, v in flat.items() if v}
    out = {"config": config, "tag": tag, "kernels": kernels, "fetch_calibration": cal, "timed_window": flat,
           "csrc_digest": buildinfo.csrc_digest()}   # bench.py quotes these counters only while the kernel sources are the ones profiled
    if flat and cal:
        print("\ntimed region of the profile run (sums over its dispatches; bench.py divides these like by like):")
        for name, r in flat.items():
            traffic = r["FETCH_SIZE_kb"] * 1024.0 * cal["factor"] + r["WRITE_SIZE_kb"] * 1024.0
            print(f"  {name:9s} iterations {r['iterations']}  dispatches {r['dispatches']}  launches with work {r['launches_with_work']}  "
                  f"traffic {traffic / 1e6:.1f} MB = {traffic / r['iterations'] / 1e6:.1f} MB per iteration  algorithmic "
                  f"{r['algorithmic_bytes'] / r['iterations'] / 1e6:.1f} MB per iteration  ratio {traffic / r['algorithmic_bytes']:.3f}"
                  + (f"  kernel time {r['duration_ns'] / r['iterations'] / 1e3:.1f} us per iteration" if r.get("duration_ns") else "")
                  + (f"  {r['fp32_flops'] / (r['duration_ns'] * 1e-9) / 1e12:.1f} TFLOP/s binary32" if r.get("duration_ns") and r.get("fp32_flops") else "")
                  + (f"  non-arithmetic VALU {r['non_arithmetic_valu_fraction']:.3f}" if "non_arithmetic_valu_fraction" in r else "")
                  + (f"  {r['valu_cycles_per_instruction']:.2f} cycles per VALU instruction" if "valu_cycles_per_instruction" in r else "")
                  + (f"  shader clock ~{r['shader_clock_mhz']:.0f} MHz" if "shader_clock_mhz" in r else ""))
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
