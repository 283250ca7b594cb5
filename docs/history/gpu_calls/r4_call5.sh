#!/bin/bash
# round 4, call 5: diagnostics of the round-3 LDS anomaly on the tree that still has it (v7); 8-class keyframe sharding tests
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_call5; mkdir -p $O
(cd _bisect/v7 && timeout 600 python diag_hb.py 2>&1 | tail -50) | tee $O/diag_hb_v7.log
echo "== keyframe shards + classes"; timeout 1200 python -m pytest tests/test_gpu_sharded_loopback.py tests/test_gpu_scale_parity.py -q -m gpu -k "keyframe_shard or keyframe_sharding or geometry_bit_exact" 2>&1 | tail -8 | tee $O/kf8.log
