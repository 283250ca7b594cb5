#!/bin/bash
# scripts/kernel_times.sh [bench args...] : rocprofv3 --kernel-trace --stats of one bench.py run on the GPU box, top kernels printed.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
rm -rf /tmp/prof_kt
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- python $R/bench.py --no-cpu-baseline --no-extras "$@" > /tmp/prof_kt.json 2>/dev/null
f=$(find /tmp/prof_kt -name "*kernel_stats.csv" | head -1)
python - "$f" <<PY
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:${TOP:-10}]:
    print("%-72s %6s  %.3f ms avg  %5s %%" % (r["Name"][:72], r["Calls"], float(r["AverageNs"])*1e-6, r["Percentage"]))
PY
