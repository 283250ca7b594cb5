#!/bin/bash
# round 6, call 3: A/B of the flavours (BENCH_ARITHMETIC now reaches bench.py) and of FTZ in the pose unit; full GPU suite, no -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r6_call3; mkdir -p $O
BENCH_ARGS="--no-extras" timeout -k 5 500 bash scripts/ab_bench.sh 3 - -:fast posenoftz:fast 2>&1 | tee $O/ab.txt
timeout -k 5 1800 python -m pytest tests -q -m gpu --durations=10 2>&1 | tail -80 > $O/gpu_tests.log
tail -40 $O/gpu_tests.log | cut -c1-400
