#!/bin/bash
# round 4, call 31: the record buffers of the intrinsics step regrow on overflow only -- the intrinsics tests
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_call31; mkdir -p $O
timeout -k 5 45 python -m pytest tests/test_gpu_intrinsics_pcg_vs_oracle.py -q -m gpu -x -k "intrinsics" 2>&1 | tail -4 | tee $O/gpu_tests.log
