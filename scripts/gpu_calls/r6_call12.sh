#!/bin/bash
# round 6, call 12: Morton reorder behind the loop's compaction: full GPU suite (the oracle applies the same rule), the drop-in call traced
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r6_call12; mkdir -p $O
timeout -k 5 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -30 > $O/gpu_tests.log
tail -4 $O/gpu_tests.log | cut -c1-300
python scripts/drop_in_profile.py 2>&1 | grep "ms per call"
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $GRAFT_REPO_ROOT/scripts/drop_in_profile.py > $O/trace.log 2>&1
grep "ms per call" $O/trace.log
python - <<'PY'
import csv, glob, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r6_call12"
f=glob.glob(O+"/trace/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:22]:
    print(f"{r['Name'][:60]:60s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:8.1f} us {float(r['TotalDurationNs'])/1e6:8.2f} ms")
import shutil; shutil.copy(f, O+"/drop_in_kernel_stats.csv")
PY
find $O/trace -name '*.csv' -size +1M -delete
