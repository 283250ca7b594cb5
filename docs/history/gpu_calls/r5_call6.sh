#!/bin/bash
# round 5, call 6: the new parity tests (row-major creation vs the unmodified reference run, the e2e golden through a TUM directory,
# MergeKeyframes), the multi-process bench tests with the rank probe, and the bench line with the replayed instrumented region
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call6; mkdir -p $O
timeout -k 5 600 python -m pytest tests/test_gpu_e2e_vga.py tests/test_gpu_tum_pipeline.py tests/test_gpu_directba_vs_oracle.py tests/test_gpu_bench_multiprocess.py tests/test_gpu_sharded_loopback.py -q -m gpu -x -s 2>&1 | grep -v "^\[bench\]\|BindScene\|amdgpu.ids" | tail -40 > $O/gpu_tests.log
tail -30 $O/gpu_tests.log | cut -c1-300
timeout -k 5 300 python bench.py --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5_call6/bench.json"))
print(d["value"], d["ms_per_step"], d.get("instrumented_region"))
print({k: d["roofline"][k] for k in ("frac","avg_launch_ms","launches","keyframes_per_launch")}, d["config"]["pose_gn_rounds_per_iteration"])
PY
tail -3 $O/bench.err
