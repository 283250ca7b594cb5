#!/bin/bash
# round 6, call 8: the whole default bench line with either flavour in the timed region (extras: intrinsics stage, PCG, cold start, drop-in)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r6_call8; mkdir -p $O
for a in exact fast; do
  BENCH_ARITHMETIC=$a timeout -k 5 400 python bench.py --no-cpu-baseline > $O/bench_$a.json 2> $O/bench_$a.log
  python - <<PY
import json
d=json.load(open("$O/bench_$a.json"))
print("$a", round(d["value"],1), "it/s  frac", round(d["roofline"]["frac"],4), " cold", round(d["cold_start"]["ba_iterations_per_s"],1), " unsorted", round(d["unsorted_ba_iterations_per_s"],1))
print("   intrinsics", d["intrinsics"]["BA_intrinsics_optimization_ms_per_iteration"], "sweep", d["intrinsics"]["sweep_ms"], " pcg", d["pcg"]["outer_iterations_per_s"], d["pcg"]["inner_steps_per_outer_iteration"], d["pcg"]["inner_steps_per_s"], " step1 ms", d.get("roofline_pcg",{}).get("avg_launch_ms"))
print("   drop_in", d["drop_in"]["ms_per_call"], d["drop_in"]["ms_per_call_iterations_only"], d["drop_in"]["lifecycle_and_end_tasks_ms_per_call"])
o=d.get("fast_math") or d.get("exact_arithmetic")
print("   other flavour:", o["arithmetic"], round(o["ba_iterations_per_s"],1), "frac", round(o["frac"],4), o["parity"])
PY
done
