#!/usr/bin/env python3
"""What rocprofv3 counts as launches of the dominant kernel against the bench line's roofline.avg_launch_ms.

Since round 4 the Gauss-Newton rounds of a pose phase are queued ahead of the host: a round queued in vain is a launch of the pose
sweep that finds no work item iterating and returns at once.  rocprofv3 --stats averages over EVERY launch; the bench line's
roofline object describes the launches that did work (the event pairs of the others are dropped, capi_ba.hip).  This script splits the
launches of the kernel trace of scripts/profile_round.sh's pass 1 by duration and writes the accounting as JSON.
usage: pose_launch_accounting.py gpurun_out/prof_<tag> profiles/<tag>_bench.json > profiles/<tag>_pose_launches.json"""
import csv
import glob
import json
import os
import sys

prof, bench_path = sys.argv[1], sys.argv[2]
line = json.load(open(bench_path))
kernel = line["roofline"]["kernel"].split("<")[0]
trace = glob.glob(os.path.join(prof, "stats", "**", "*kernel_trace.csv"), recursive=True)[0]
durations = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(trace)) if kernel in r["Kernel_Name"]]
in_vain = [d for d in durations if d < 10.0]            # a launch that returns at once takes 3-5 us
worked = [d for d in durations if d >= 10.0]
first = [d for d in worked if d >= 0.5 * max(worked)]   # first rounds of a phase: the whole cloud against every non-converged keyframe
later = [d for d in worked if d < 0.5 * max(worked)]
mean = lambda v: sum(v) / len(v) if v else 0.0
json.dump({"kernel": line["roofline"]["kernel"], "trace": "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extras",
           "all_launches": {"n": len(durations), "avg_us": mean(durations)},
           "queued_in_vain": {"n": len(in_vain), "avg_us": mean(in_vain)},
           "worked": {"n": len(worked), "avg_us": mean(worked), "first_rounds": {"n": len(first), "avg_us": mean(first)},
                      "later_rounds": {"n": len(later), "avg_us": mean(later)}},
           "bench_line": {"file": os.path.basename(bench_path), "roofline_avg_launch_us": 1e3 * line["roofline"]["avg_launch_ms"],
                          "launches": line["roofline"]["launches"]}}, sys.stdout, indent=1)
