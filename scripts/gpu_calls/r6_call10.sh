#!/bin/bash
# round 6, call 10: pipelined merge batch after the packing-slot fix; kernel trace of the drop-in call (where do the 53 ms go now?)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r6_call10; mkdir -p $O
timeout -k 5 600 python -m pytest tests/test_gpu_lifecycle_stages.py -q -m gpu 2>&1 | tail -4
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $GRAFT_REPO_ROOT/scripts/drop_in_profile.py > $O/trace.log 2>&1
grep "ms per call" $O/trace.log
python - <<'PY'
import csv, glob, os, collections
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r6_call10"
f=glob.glob(O+"/trace/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:16]:
    print(f"{r['Name'][:60]:60s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:8.1f} us {float(r['TotalDurationNs'])/1e6:8.2f} ms")
import shutil; shutil.copy(f, O+"/drop_in_kernel_stats.csv")
t=glob.glob(O+"/trace/**/*kernel_trace.csv", recursive=True)[0]
rows=list(csv.DictReader(open(t)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the last call: find the last 'delete_update' and print gap statistics by kernel over the preceding 60 ms
end=int(rows[-1]['End_Timestamp'])
sel=[r for r in rows if int(r['Start_Timestamp'])>end-55_000_000]
gaps=collections.defaultdict(list); prev=None
for r in sel:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    if prev is not None: gaps[r['Kernel_Name'].split('(')[0][-40:]].append((s-prev)/1e3)
    prev=e
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in sel)/1e6
print(f"last 55 ms: {len(sel)} dispatches, kernel time {busy:.1f} ms, idle {55-busy:.1f} ms")
for k,v in sorted(gaps.items(), key=lambda kv:-sum(kv[1]))[:12]:
    print(f"  gap before {k:40s} n={len(v):5d} mean {sum(v)/len(v):6.1f} us total {sum(v)/1e3:6.2f} ms")
with open(O+"/merge_window.txt","w") as out:
    idx=[i for i,r in enumerate(sel) if 'merge_decide' in r['Kernel_Name']]
    if idx:
        i0=idx[len(idx)//2]; prev=int(sel[i0-1]['End_Timestamp'])
        for r in sel[i0:i0+24]:
            s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
            line=f"{(s-prev)/1e3:7.1f} us gap | {(e-s)/1e3:7.1f} us | {r['Kernel_Name'].split('(')[0][-50:]}"
            print(line); out.write(line+"\n"); prev=e
PY
find $O/trace -name '*.csv' -size +1M -delete
