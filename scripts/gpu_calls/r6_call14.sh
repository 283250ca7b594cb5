#!/bin/bash
# round 6, call 14: merge batch by cell lists (one launch per keyframe): lifecycle tests, the whole suite, drop-in with and without
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r6_call14; mkdir -p $O
timeout -k 5 600 python -m pytest tests/test_gpu_lifecycle_stages.py -q -m gpu -x 2>&1 | tail -30 | cut -c1-300
timeout -k 5 1500 python -m pytest tests -q -m gpu 2>&1 | tail -30 > $O/gpu_tests.log
tail -12 $O/gpu_tests.log | cut -c1-300
python scripts/drop_in_profile.py 2>&1 | grep "ms per call"
BAHIP_MERGE_CELLS=0 python scripts/drop_in_profile.py 2>&1 | grep "ms per call"
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $GRAFT_REPO_ROOT/scripts/drop_in_profile.py > $O/trace.log 2>&1
grep "ms per call" $O/trace.log
python - <<'PY'
import csv, glob, os, collections
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r6_call14"
f=glob.glob(O+"/trace/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:24]:
    print(f"{r['Name'][:60]:60s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:8.1f} us {float(r['TotalDurationNs'])/1e6:8.2f} ms")
import shutil; shutil.copy(f, O+"/drop_in_kernel_stats.csv")
t=glob.glob(O+"/trace/**/*kernel_trace.csv", recursive=True)[0]
rows=list(csv.DictReader(open(t)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
end=int(rows[-1]['End_Timestamp'])
W=40_000_000
sel=[r for r in rows if int(r['Start_Timestamp'])>end-W]
gaps=collections.defaultdict(list); byk=collections.defaultdict(float); prev=None
for r in sel:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    name=r['Kernel_Name'].split('(')[0][-40:]
    byk[name]+=(e-s)/1e6
    if prev is not None: gaps[name].append((s-prev)/1e3)
    prev=e
busy=sum(byk.values())
print(f"last {W/1e6:.0f} ms: {len(sel)} dispatches, kernel time {busy:.1f} ms")
for k,v in sorted(byk.items(), key=lambda kv:-kv[1])[:14]: print(f"  {k:42s} {v:6.2f} ms")
for k,v in sorted(gaps.items(), key=lambda kv:-sum(kv[1]))[:8]:
    print(f"  gap before {k:40s} n={len(v):5d} mean {sum(v)/len(v):6.1f} us total {sum(v)/1e3:6.2f} ms")
PY
find $O/trace -name '*.csv' -size +1M -delete
