#!/bin/bash
# round 6, call 2: the two arithmetic flavours in one library -- full GPU suite (exact parity after the split of the kernel units, the
# fast flavour's tolerance tests), A/B of the flavours and of FTZ in the pose unit
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r6_call2; mkdir -p $O
BENCH_ARGS="--no-extras" timeout -k 5 500 bash scripts/ab_bench.sh 3 - -:fast posenoftz:fast 2>&1 | tee $O/ab.txt
timeout -k 5 1500 python -m pytest tests -q -m gpu -x --durations=15 2>&1 | tail -60 > $O/gpu_tests.log
tail -45 $O/gpu_tests.log | cut -c1-300
