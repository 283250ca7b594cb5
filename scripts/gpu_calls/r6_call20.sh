#!/bin/bash
# round 6, call 20: what separates dependent launches (scripts/microbench/launch_gap.hip), and the gaps of the device-driven loop itself
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r6_call20; mkdir -p $O
hipcc --offload-arch=gfx950 -O2 -o /tmp/launch_gap scripts/microbench/launch_gap.hip
cd /tmp
timeout -k 5 120 rocprofv3 --kernel-trace --output-format csv -d $O/gap -- /tmp/launch_gap > $O/gap.log 2>&1
python $GRAFT_REPO_ROOT/scripts/microbench/launch_gap.py $O/gap | tee $O/launch_gap.txt
# the device-driven loop on the whole cloud and on an eighth of it: gaps by kernel over the timed region
for W in 1 8; do
  EXTRA=""; [ $W = 8 ] && EXTRA="--emulate-world 8 --force-allreduce"
  timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $O/loop$W -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras $EXTRA > $O/bench_loop$W.json 2> $O/bench_loop$W.log
  python - <<PY
import csv, glob, collections, json
t=glob.glob("$O/loop$W/**/*kernel_trace.csv", recursive=True)[0]
rows=sorted(csv.DictReader(open(t)), key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "iteration_begin_kernel" in r["Kernel_Name"]]
print("world $W:", json.load(open("$O/bench_loop$W.json"))["ms_per_step"], "ms per iteration in the bench line;", len(idx), "iterations in the trace")
# bench.py: pre-pass 3 + 20, warm-up 3, the timed region 20 (no event records), then the instrumented replay 3 + 20 (event pairs around the pose launches)
for name,(a,b) in {"timed region":(26,46),"instrumented replay":(49,69)}.items():
    if b>=len(idx): continue
    sel=rows[idx[a]:idx[b]]
    span=(int(rows[idx[b]]["Start_Timestamp"])-int(sel[0]["Start_Timestamp"]))/1e3/20
    gap=collections.defaultdict(list); dur=collections.defaultdict(list); prev=None
    for r in sel:
        s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"]); n=r["Kernel_Name"].split("(")[0][-44:]
        if prev is not None: gap[n].append((s-prev)/1e3)
        dur[n].append((e-s)/1e3); prev=e
    print(f"  {name}: {span:.1f} us per iteration, {len(sel)} dispatches")
    for n in dur: print(f"     {n:46s} n={len(dur[n]):3d} dur {sum(dur[n])/len(dur[n]):7.1f} us  gap before {sum(gap[n])/max(1,len(gap[n])):5.1f} us")
PY
done 2>&1 | tee $O/loop_gaps.txt
find $O -name '*kernel_trace.csv' -size +1M -delete
