#!/usr/bin/env python3
"""Mean gap (previous kernel's end -> this kernel's start) and duration per kernel of scripts/microbench/launch_gap.hip from a rocprofv3 kernel trace."""
import collections, csv, glob, sys
t = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(t)), key=lambda r: int(r["Start_Timestamp"]))
gap, dur, prev = collections.defaultdict(list), collections.defaultdict(list), None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0]
    if prev is not None and prev[1] == name or (prev is not None and name == "after_writer_kernel"):
        gap[name].append((s - prev[0]) / 1e3)
    dur[name].append((e - s) / 1e3)
    prev = (e, name if name != "writer_kernel" else "after_writer_kernel")
for name in dur:
    g = sorted(gap.get(name, [0.0]))
    print(f"{name:24s} n={len(dur[name]):4d}  duration {sum(dur[name]) / len(dur[name]):8.2f} us   gap before: median {g[len(g) // 2]:6.2f} us  mean {sum(g) / len(g):6.2f} us")
