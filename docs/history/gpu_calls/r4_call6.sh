#!/bin/bash
# round 4, call 6: the LDS anomaly on the tree that still has it -- v8 per-lane row offset, v9 scalar copy of the item, v11 today's
# sink shape (32-bit LDS address, address_space(3) atomic) transplanted; each with the K = 200 test and the table diagnostics
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_call6; mkdir -p $O
for v in v8 v9 v11; do
  echo "== bisect $v"
  (cd _bisect/$v && timeout 600 python -m pytest tests/test_gpu_scale_parity.py -q -m gpu -k "batched_pose_estimation and lds" 2>&1 | tail -3; timeout 300 python diag_hb.py 2>&1 | head -3) | tee $O/bisect_$v.log
done
