#!/bin/bash
# round 5, call 7: lifecycle with the fused append (scan + append + advance in one launch) and without the per-keyframe fill of a merge
# batch: the bit-exact lifecycle / e2e / DirectBA tests, then the bench line with its extras (drop_in)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call7; mkdir -p $O
timeout -k 5 600 python -m pytest tests/test_gpu_lifecycle_stages.py tests/test_gpu_e2e_vga.py tests/test_gpu_directba_vs_oracle.py tests/test_gpu_golden_reference.py tests/test_gpu_edge_cases.py tests/test_gpu_directba_cpp.py tests/test_gpu_kernels_vs_oracle.py -q -m gpu -x 2>&1 | tail -15 > $O/gpu_tests.log
tail -6 $O/gpu_tests.log | cut -c1-300
timeout -k 5 400 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5_call7/bench.json"))
print(d["value"], d["ms_per_step"], d["instrumented_region"]["slowdown_by_event_records"])
print("drop_in", d.get("drop_in"))
print("cold", d.get("cold_start",{}).get("ba_iterations_per_s"), "unsorted", d.get("unsorted_ba_iterations_per_s"), "pcg", d.get("pcg",{}).get("outer_iterations_per_s"), "intr", d.get("intrinsics"))
PY
tail -3 $O/bench.err
