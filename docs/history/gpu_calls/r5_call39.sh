#!/bin/bash
# round 5, call 39: full GPU suite + smoke + the default bench line on the round's last code state (after the creation-batch change)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call39; mkdir -p $O
timeout -k 5 700 python -m pytest tests -q -m gpu 2>&1 | tail -6 > $O/gpu_tests.log
tail -3 $O/gpu_tests.log | cut -c1-300
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
export BADSLAM_RENDER_WORKERS=32
timeout -k 5 500 python bench.py > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5_call39/bench.json"))
print(d["value"], d["ms_per_step"], d["instrumented_region"]["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["loop"]["timed_calls_driven_by_the_device"], d["prepass"]["iterations"])
print("traffic", d["roofline"].get("traffic_source"), d["roofline"].get("traffic_over_algorithmic"), d["roofline_geometry"].get("traffic_over_algorithmic"), d["stage_ms_per_iteration"])
print("drop_in", d["drop_in"]["ms_per_call"], d["drop_in"]["ba_iterations_per_s"], d["drop_in"]["ms_per_call_iterations_only"], "cold", d["cold_start"]["ba_iterations_per_s"], "unsorted", d["unsorted_ba_iterations_per_s"])
print("pcg", d["pcg"]["outer_iterations_per_s"], d["pcg"]["inner_steps_per_outer_iteration"], d["pcg"]["inner_steps_per_s"], "intr", d["intrinsics"]["BA_intrinsics_optimization_ms_per_iteration"])
print("cpu", d["cpu_baseline"]["seconds_per_cost_evaluation"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["kind"], d["cpu_baseline"].get("port",{}).get("seconds_per_cost_evaluation"))
PY
