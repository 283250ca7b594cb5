#!/bin/bash
# round 6, call 39: kernel by kernel, the TIMED region of the emulated 8-rank share (no pre-pass: the timed region is iterations 3 .. 22 of the trace)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=8
O=$GRAFT_REPO_ROOT/gpurun_out/r6_call39; mkdir -p $O
cd /tmp
for W in 8 1; do
  EXTRA=""; [ $W = 8 ] && EXTRA="--emulate-world 8 --force-allreduce"
  timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $O/loop$W -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-prepass $EXTRA > $O/bench_loop$W.json 2> $O/bench_loop$W.log
  python - <<PY
import csv, glob, collections, json
t=glob.glob("$O/loop$W/**/*kernel_trace.csv", recursive=True)[0]
rows=sorted(csv.DictReader(open(t)), key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "iteration_begin_kernel" in r["Kernel_Name"]]
print("world $W:", json.load(open("$O/bench_loop$W.json"))["ms_per_step"], "ms per iteration in the bench line;", len(idx), "iterations in the trace")
a,b=3,23
sel=rows[idx[a]:idx[b]]
span=(int(rows[idx[b]]["Start_Timestamp"])-int(sel[0]["Start_Timestamp"]))/1e3/20
gap=collections.defaultdict(list); dur=collections.defaultdict(list); prev=None
for r in sel:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"]); n=r["Kernel_Name"].split("(")[0][-46:]
    if prev is not None: gap[n].append((s-prev)/1e3)
    dur[n].append((e-s)/1e3); prev=e
print(f"  timed region: {span:.1f} us per iteration, {len(sel)} dispatches")
for n in dur: print(f"     {n:48s} n={len(dur[n]):3d} per iteration {sum(dur[n])/20:7.1f} us (avg {sum(dur[n])/len(dur[n]):6.1f})  gaps {sum(gap[n])/20:5.1f} us")
acc=[d for n in dur if "pose_accumulate" in n for d in dur[n]]
print("     pose launches by duration:", sorted(round(x) for x in acc)[:12], "...", sorted(round(x) for x in acc)[-5:])
PY
done 2>&1 | tee $O/timed_region.txt
find $O -name '*kernel_trace.csv' -size +1M -delete
