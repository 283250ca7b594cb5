#!/bin/bash
# round 4, call 13: on the first commit where the immediate LDS adds are right (b4280f8), take back its batch protocol (R1) / its tile
# schedule lookup in the LDS kernel (R2): which one brings the wrong sums back?  (60 s limits: a protocol edit hung a box once)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_call13; mkdir -p $O
for v in R1 R2; do
  echo "== $v"
  (cd _bisect/$v && timeout 60 python -m pytest tests/test_gpu_scale_parity.py -q -m gpu -k "batched_pose_estimation and lds" 2>&1 | tail -2; echo "exit $?") | tee -a $O/revert.log
done
