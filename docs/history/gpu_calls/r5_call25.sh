#!/bin/bash
# round 5, call 25: timeline of the drop-in calls (is the GPU busy throughout, or waiting for the host's launches?)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=8
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call25; mkdir -p $O
cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $GRAFT_REPO_ROOT/scripts/drop_in_profile.py > $O/trace.log 2>&1
grep drop_in $O/trace.log | tail -3
f=$(find $O/trace -name '*kernel_trace.csv' | xargs ls -S | head -1)
python - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the three timed calls: everything after the last-but-3 ... approximate by taking the last 60 % of the lifecycle launches; simpler: the
# window from the 4th-last delete_update kernel (one per call's end tasks) to the end
ends = [i for i, r in enumerate(rows) if "delete_update_kernel" in r["Kernel_Name"]]
print("delete_update launches", len(ends))
a = ends[-4] + 1 if len(ends) >= 4 else 0
w = rows[a:]
t0, t1 = int(w[0]["Start_Timestamp"]), int(w[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in w)
gaps = [(int(w[i]["Start_Timestamp"]) - int(w[i-1]["End_Timestamp"])) / 1e3 for i in range(1, len(w))]
import collections
print("window ms", (t1 - t0) / 1e6, "kernel ms", busy / 1e6, "launches", len(w), "busy fraction", busy / (t1 - t0))
hist = collections.Counter(min(int(g // 2) * 2, 40) for g in gaps if g > 0)
print("gap histogram (us bucket: count):", sorted(hist.items()))
print("sum of gaps > 2 us (ms):", sum(g for g in gaps if g > 2) / 1e3, " > 20 us:", sum(g for g in gaps if g > 20) / 1e3)
by = collections.Counter()
for i in range(1, len(w)):
    if gaps[i-1] > 2: by[w[i]["Kernel_Name"].split("(")[0][-40:]] += gaps[i-1]
for k, v in by.most_common(12): print("  gap before %-42s %8.2f ms" % (k, v / 1e3))
PY
rm -rf $O/trace
