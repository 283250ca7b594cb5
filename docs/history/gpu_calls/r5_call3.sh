#!/bin/bash
# round 5, call 3: instruction costs (more classes), the limb test with the branch-free split, single-bit A/B of the trims
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call3; mkdir -p $O
timeout -k 3 60 scripts/experiments/bin/inst_cost 4 > $O/inst_cost.txt 2>&1
tail -32 $O/inst_cost.txt
timeout -k 5 200 python -m pytest tests/test_gpu_kernels_vs_oracle.py -q -m gpu -x 2>&1 | tail -3 | tee $O/gpu_tests.log
BENCH_ARGS="--no-extras" timeout -k 5 500 bash scripts/ab_bench.sh 3 trim0 trim1 trim2 trim4 trim8 trim16 trim96 - 2>&1 | tee $O/ab.txt
