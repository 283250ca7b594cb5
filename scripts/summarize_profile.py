#!/usr/bin/env python3
"""Condenses a scripts/profile_round.sh output directory: per-kernel time from rocprofv3's
kernel_stats.csv and per-kernel, per-launch FETCH_SIZE / WRITE_SIZE averages from the PMC passes.
Writes <dir>/kernel_stats.csv (copy), <dir>/pmc_per_kernel.json and prints a table."""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def short(name):
    return name.split("(")[0].replace("bahip::", "").replace("(anonymous namespace)::", "").replace("void ", "")[:70]


def main():
    d = sys.argv[1]
    stats = glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True)
    rows = []
    if stats:
        shutil.copy(stats[0], os.path.join(d, "kernel_stats.csv"))
        with open(stats[0]) as f:
            rows = list(csv.DictReader(f))
        print(f"{'kernel':72s} {'calls':>7s} {'avg us':>10s} {'total ms':>10s} {'%':>6s}")
        for r in rows[:16]:
            print(f"{short(r['Name']):72s} {int(r['Calls']):7d} {float(r['AverageNs']) / 1e3:10.1f} "
                  f"{float(r['TotalDurationNs']) / 1e6:10.2f} {float(r['Percentage']):6.2f}")
    pmc = {}
    for sub in ("pmc_fetch", "pmc_write"):
        for path in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
            acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
            with open(path) as f:
                for r in csv.DictReader(f):
                    a = acc[short(r["Kernel_Name"])][r["Counter_Name"]]
                    a[0] += float(r["Counter_Value"])
                    a[1] += 1
            for k, counters in acc.items():
                for c, (total, n) in counters.items():
                    pmc.setdefault(k, {})[c] = {"avg_per_launch": total / n, "launches": n}
    if pmc:
        with open(os.path.join(d, "pmc_per_kernel.json"), "w") as f:
            json.dump(pmc, f, indent=1, sort_keys=True)
        print("\nPMC per launch (rocprofv3 units: FETCH_SIZE / WRITE_SIZE in KB, uncorrected):")
        for k in sorted(pmc, key=lambda k: -pmc[k].get("FETCH_SIZE", {"avg_per_launch": 0})["avg_per_launch"])[:12]:
            print(f"{k:72s} " + "  ".join(f"{c}={v['avg_per_launch']:.1f} (n={v['launches']})" for c, v in sorted(pmc[k].items())))


if __name__ == "__main__":
    main()
