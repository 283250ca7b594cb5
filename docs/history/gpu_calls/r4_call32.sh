#!/bin/bash
# round 4, call 32: the 200-keyframe intrinsics step (several chunks and slices per record buffer) after the regrow change
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_call32; mkdir -p $O
timeout -k 3 26 python -m pytest tests/test_gpu_scale_parity.py -q -m gpu -x -k "intrinsics_step" 2>&1 | tail -3 | tee $O/gpu_tests.log
