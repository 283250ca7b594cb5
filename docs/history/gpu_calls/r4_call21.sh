#!/bin/bash
# round 4, call 21: device-side creation batch -- lifecycle / end-to-end / sharded parity, then the bench line (drop_in)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_call21; mkdir -p $O
timeout -k 5 300 python -m pytest tests/test_gpu_lifecycle_stages.py tests/test_gpu_e2e_vga.py tests/test_gpu_directba_vs_oracle.py tests/test_gpu_directba_cpp.py tests/test_gpu_edge_cases.py tests/test_gpu_sharded_loopback.py -q -m gpu -x 2>&1 | tail -25 > $O/gpu_tests.log
tail -4 $O/gpu_tests.log
timeout -k 5 120 python -m pytest tests/test_gpu_scale_parity.py -q -m gpu -x -k "c2 or creation" 2>&1 | tail -4 >> $O/gpu_tests.log
tail -2 $O/gpu_tests.log
timeout -k 5 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4_call21/bench.json"))
print(round(d["value"],1), d["drop_in"]["ms_per_call"], d["drop_in"]["lifecycle_and_end_tasks_ms_per_call"], d["drop_in"]["surfels"], d["intrinsics"], d["pcg"]["outer_iterations_per_s"])
PY
