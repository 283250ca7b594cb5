#!/bin/bash
# round 5, call 31 (call 21 again, after the two-stage normals pass and the LDS sums of the intrinsics sweep): artifact set r5_e of the final tree (device-driven loop restored, pre-pass in the bench): full GPU suite, the profile
# passes, then the default bench line reading its own counters, the launch accounting, and the emulated shares of a surfel-sharded run
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call31; mkdir -p $O
timeout -k 5 700 python -m pytest tests -q -m gpu 2>&1 | tail -6 > $O/gpu_tests.log
tail -3 $O/gpu_tests.log | cut -c1-300
export BADSLAM_RENDER_WORKERS=32
PASS_TIMEOUT=150 bash scripts/profile_round.sh r5_e > $O/profile_r5_e.log 2>&1
grep -A3 "timed region of the profile" $O/profile_r5_e.log | cut -c1-600
P=gpurun_out/prof_r5_e
cp $P/kernel_stats.csv profiles/r5_e_kernel_stats.csv; cp $P/pmc_per_kernel.json profiles/r5_e_pmc_per_kernel.json; cp $P/summary.txt profiles/r5_e_summary.txt
timeout -k 5 500 python bench.py > $O/bench.json 2> $O/bench.err
cp $O/bench.json profiles/r5_e_bench.json
python scripts/pose_launch_accounting.py $P profiles/r5_e_bench.json > $O/pose_launches.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5_call31/bench.json"))
print(d["value"], d["ms_per_step"], d["instrumented_region"]["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["loop"]["timed_calls_driven_by_the_device"], d["prepass"]["iterations"])
print("traffic", d["roofline"].get("traffic_source"), d["roofline"].get("traffic_over_algorithmic"), d["roofline_geometry"].get("traffic_over_algorithmic"))
print("drop_in", d["drop_in"]["ms_per_call"], d["drop_in"]["ba_iterations_per_s"], "cold", d["cold_start"]["ba_iterations_per_s"], "unsorted", d["unsorted_ba_iterations_per_s"])
print("pcg", d["pcg"]["outer_iterations_per_s"], d["pcg"]["inner_steps_per_outer_iteration"], d["pcg"]["inner_steps_per_s"], "intr", d["intrinsics"]["BA_intrinsics_optimization_ms_per_iteration"])
print("cpu", d["cpu_baseline"]["seconds_per_cost_evaluation"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["kind"], d["cpu_baseline"].get("port",{}).get("seconds_per_cost_evaluation"))
PY
for w in 8 4 2; do
  timeout -k 5 100 python bench.py --emulate-world $w --force-allreduce --no-cpu-baseline --no-extras > $O/bench_emu$w.json 2> $O/bench_emu$w.err
  python -c "
import json; d=json.load(open('$O/bench_emu$w.json')); print('emulated world $w:', round(d['ms_per_step'],4), 'ms per iteration', d['loop'])"
done
