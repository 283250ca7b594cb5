#!/bin/bash
# round 6, call 19: visible-tile lists in one pass; full GPU suite; drop-in traced with the launch count of one call
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r6_call19; mkdir -p $O
timeout -k 5 1500 python -m pytest tests -q -m gpu 2>&1 | tail -30 > $O/gpu_tests.log
tail -6 $O/gpu_tests.log | cut -c1-300
python scripts/drop_in_profile.py 2>&1 | grep "ms per call"
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $GRAFT_REPO_ROOT/scripts/drop_in_profile.py > $O/trace.log 2>&1
grep "ms per call" $O/trace.log
python - <<'PY'
import csv, glob, os, collections
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r6_call19"
f=glob.glob(O+"/trace/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:14]:
    print(f"{r['Name'][:60]:60s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:8.1f} us {float(r['TotalDurationNs'])/1e6:8.2f} ms")
import shutil; shutil.copy(f, O+"/drop_in_kernel_stats.csv")
t=glob.glob(O+"/trace/**/*kernel_trace.csv", recursive=True)[0]
rows=list(csv.DictReader(open(t)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the last call = from the last lifecycle_bounds_kernel-before-create (third from the end: creation, merge, end-task merge) to the end
idx=[i for i,r in enumerate(rows) if 'lifecycle_bounds_kernel' in r['Kernel_Name']]
start=idx[-3]
sel=rows[start:]
byk=collections.Counter(); tk=collections.defaultdict(float)
for r in sel:
    n=r['Kernel_Name'].split('(')[0][-44:]; byk[n]+=1; tk[n]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
span=(int(sel[-1]['End_Timestamp'])-int(sel[0]['Start_Timestamp']))/1e6
with open(O+"/last_call_launches.txt","w") as out:
    print(f"last call (first lifecycle_bounds_kernel of the creation batch .. end): {len(sel)} dispatches in {span:.1f} ms, kernel time {sum(tk.values()):.1f} ms", file=out)
    for n,c in byk.most_common(40): print(f"  {n:46s} {c:5d} launches {tk[n]:7.2f} ms", file=out)
print(open(O+"/last_call_launches.txt").read())
PY
find $O/trace -name '*.csv' -size +1M -delete
