#!/bin/bash
# round 5, call 11: after the split of the C boundary into six translation units: the full GPU suite, the default bench line (with the
# CPU baseline and the extras), the profile passes of the same command
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call11; mkdir -p $O
timeout -k 5 600 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $O/gpu_tests.log
tail -4 $O/gpu_tests.log | cut -c1-300
timeout -k 5 500 python bench.py > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5_call11/bench.json"))
print(d["value"], d["ms_per_step"], d["instrumented_region"]["slowdown_by_event_records"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
print("drop_in", d["drop_in"]["ms_per_call"], d["drop_in"]["ba_iterations_per_s"], "cold", d["cold_start"]["ba_iterations_per_s"], "unsorted", d["unsorted_ba_iterations_per_s"])
print("pcg", d["pcg"]["outer_iterations_per_s"], d["pcg"]["inner_steps_per_outer_iteration"], "intr", d["intrinsics"]["BA_intrinsics_optimization_ms_per_iteration"])
print("cpu", d["cpu_baseline"]["seconds_per_cost_evaluation"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["kind"], d["cpu_baseline"].get("port",{}).get("seconds_per_cost_evaluation"))
PY
export BADSLAM_RENDER_WORKERS=32
PASS_TIMEOUT=150 bash scripts/profile_round.sh r5_b > $O/profile_r5_b.log 2>&1
grep -A3 "timed region of the profile" $O/profile_r5_b.log | cut -c1-600
