#!/bin/bash
# round 5, call 40: the persistent pose sweep's batch size (positions a workgroup draws from its XCD's queue per global atomic): 32 (built)
# against 64 and 128, alternating
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call40; mkdir -p $O
BENCH_ARGS="--no-extras" bash scripts/ab_bench.sh 3 - batch64 batch128 2>&1 | tee $O/ab.txt
