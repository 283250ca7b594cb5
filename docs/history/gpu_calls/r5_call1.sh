#!/bin/bash
# round 5, call 1: instruction-cost microbenchmark (issue cost per class + the shader clock it runs at) and the baseline bench of HEAD
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call1; mkdir -p $O
timeout -k 3 120 scripts/experiments/bin/inst_cost 1 2 4 8 > $O/inst_cost.txt 2>&1
grep -A40 "4 wavefronts" $O/inst_cost.txt | head -34
BENCH_ARGS="--no-extras" timeout -k 5 200 bash scripts/ab_bench.sh 2 - 2>&1 | tee $O/ab.txt
