#!/bin/bash
# round 6, call 16: merge_pairs with its rows requested up front: lifecycle tests, drop-in, kernel stats
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/${CALL:-r6_call16}; mkdir -p $O
timeout -k 5 600 python -m pytest tests/test_gpu_lifecycle_stages.py tests/test_gpu_directba_vs_oracle.py -q -m gpu -x 2>&1 | tail -5 | cut -c1-300
python scripts/drop_in_profile.py 2>&1 | grep "ms per call"
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $GRAFT_REPO_ROOT/scripts/drop_in_profile.py > $O/trace.log 2>&1
grep "ms per call" $O/trace.log
python - <<'PY'
import csv, glob, os, collections
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/"+os.environ.get("CALL","r6_call16")
f=glob.glob(O+"/trace/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:16]:
    print(f"{r['Name'][:60]:60s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:8.1f} us {float(r['TotalDurationNs'])/1e6:8.2f} ms")
import shutil; shutil.copy(f, O+"/drop_in_kernel_stats.csv")
PY
find $O/trace -name '*.csv' -size +1M -delete
