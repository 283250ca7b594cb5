#!/bin/bash
# round 5, call 2: parity of the trimmed sweeps (BAHIP_TRIM = 127) + A/B against the round-4 spellings on one box + instruction costs
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call2; mkdir -p $O
timeout -k 3 60 scripts/experiments/bin/inst_cost 4 1 > $O/inst_cost.txt 2>&1
head -34 $O/inst_cost.txt
timeout -k 5 300 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $O/gpu_tests.log
tail -5 $O/gpu_tests.log
BENCH_ARGS="--no-extras" timeout -k 5 400 bash scripts/ab_bench.sh 3 trim0 - trim95 trim63 2>&1 | tee $O/ab.txt
