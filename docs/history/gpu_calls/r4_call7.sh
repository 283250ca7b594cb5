#!/bin/bash
# round 4, call 7: the commit that fixed the round-3 LDS anomaly as it is (v12) and with its adds placed right behind the reduction again (v13)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_call7; mkdir -p $O
for v in v12 v13; do
  echo "== bisect $v"
  (cd _bisect/$v && timeout 600 python -m pytest tests/test_gpu_scale_parity.py -q -m gpu -k "batched_pose_estimation and lds" 2>&1 | tail -3; timeout 300 python diag_hb.py 2>&1 | head -4) | tee $O/bisect_$v.log
done
