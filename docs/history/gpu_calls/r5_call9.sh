#!/bin/bash
# round 5, call 9: full GPU suite (mask replay in the geometry step, tagged-word fused append), A/B of the mask replay, drop-in timing
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call9; mkdir -p $O
timeout -k 5 600 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $O/gpu_tests.log
tail -4 $O/gpu_tests.log | cut -c1-300
BENCH_ARGS="--no-extras" timeout -k 5 300 bash scripts/ab_bench.sh 3 nomask - 2>&1 | tee $O/ab.txt
python scripts/drop_in_profile.py 2>&1 | grep "ms per call"
cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $GRAFT_REPO_ROOT/scripts/drop_in_profile.py > $O/trace.log 2>&1
python - <<'PY'
import csv, glob, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5_call9"
f=glob.glob(O+"/trace/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:14]:
    print(f"{r['Name'][:70]:70s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:8.1f} us {float(r['TotalDurationNs'])/1e6:8.2f} ms")
import shutil; shutil.copy(f, O+"/drop_in_kernel_stats.csv")
PY
find $O/trace -name '*kernel_trace.csv' -size +2M -delete
