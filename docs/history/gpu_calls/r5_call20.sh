#!/bin/bash
# round 5, call 20: the sweeps' run order is rebuilt after the backend (or an upload) rearranged the surfels (it used to go stale for up to
# 32 pose phases): schedule / lifecycle / loop tests, then the bench line with its extras (unsorted, drop_in, cold start)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call20; mkdir -p $O
timeout -k 5 600 python -m pytest tests/test_gpu_tile_schedule.py tests/test_gpu_lifecycle_stages.py tests/test_gpu_device_loop.py tests/test_gpu_e2e_vga.py tests/test_gpu_directba_vs_oracle.py tests/test_gpu_edge_cases.py -q -m gpu -x 2>&1 | tail -4 | tee $O/gpu_tests.log | cut -c1-300
for r in 1 2; do
timeout -k 5 400 python bench.py --no-cpu-baseline > $O/bench_$r.json 2> $O/bench_$r.err
python - $r <<'PY'
import json, sys
d=json.load(open("gpurun_out/r5_call20/bench_%s.json" % sys.argv[1]))
print(d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "launch", d["roofline"]["avg_launch_ms"])
print("drop_in", d["drop_in"]["ms_per_call"], "iterations only", d["drop_in"]["ms_per_call_iterations_only"], "cold", d["cold_start"]["ba_iterations_per_s"], "unsorted", d["unsorted_ba_iterations_per_s"])
print("pcg", d["pcg"]["outer_iterations_per_s"], d["pcg"]["inner_steps_per_outer_iteration"], "intr", d["intrinsics"]["BA_intrinsics_optimization_ms_per_iteration"])
PY
done
