#!/bin/bash
# round 5, call 5: GPU suite (rank probe, PCG flag in exchange 2, NaN-on-one-rank test, schedule under sharding), then the round's
# profile passes with the per-dispatch window accounting
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call5; mkdir -p $O
timeout -k 5 420 python -m pytest tests -q -m gpu -x 2>&1 | tail -25 > $O/gpu_tests.log
tail -6 $O/gpu_tests.log
PASS_TIMEOUT=120 bash scripts/profile_round.sh r5_a > $O/profile_r5_a.log 2>&1
tail -12 $O/profile_r5_a.log
