#!/bin/bash
# round 4, call 9: from which commit on does the IMMEDIATE form of the LDS adds give the right sums?  (each tree = that commit with the
# one-line change that puts the adds right behind the reduction)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_call9; mkdir -p $O
for c in b73d4ca 751969f 58a14bc 45af264 b4280f8 e9feb0c 99eb6a9 fea30eb; do
  echo "== immediate adds on $c"
  (cd _bisect/c_$c && timeout 600 python -m pytest tests/test_gpu_scale_parity.py -q -m gpu -k "batched_pose_estimation and lds" 2>&1 | tail -2) | tee -a $O/bisect_commits.log
done
