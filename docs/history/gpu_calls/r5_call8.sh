#!/bin/bash
# round 5, call 8: lifecycle tests again (planes emptied only inside frame-aware batches), then a kernel trace of the drop-in calls
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call8; mkdir -p $O
timeout -k 5 600 python -m pytest tests/test_gpu_lifecycle_stages.py tests/test_gpu_e2e_vga.py tests/test_gpu_directba_vs_oracle.py tests/test_gpu_golden_reference.py tests/test_gpu_edge_cases.py tests/test_gpu_directba_cpp.py tests/test_gpu_kernels_vs_oracle.py -q -m gpu -x 2>&1 | tail -15 > $O/gpu_tests.log
tail -4 $O/gpu_tests.log | cut -c1-300
python scripts/drop_in_profile.py 2>&1 | grep drop_in
cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $GRAFT_REPO_ROOT/scripts/drop_in_profile.py > $O/trace.log 2>&1
grep drop_in $O/trace.log
python - <<'PY'
import csv, glob, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5_call8"
f=glob.glob(O+"/trace/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:26]:
    print(f"{r['Name'][:70]:70s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:8.1f} us {float(r['TotalDurationNs'])/1e6:8.2f} ms")
import shutil; shutil.copy(f, O+"/drop_in_kernel_stats.csv")
PY
find $O/trace -name '*kernel_trace.csv' -size +2M -delete
