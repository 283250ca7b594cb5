#!/bin/bash
# round 6, call 29: hand-over of a call with surfel updates to the device loop: the tests, the whole suite, the drop-in call with and without
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r6_call29; mkdir -p $O
timeout -k 5 600 python -m pytest tests/test_gpu_directba_vs_oracle.py tests/test_gpu_device_loop.py -q -m gpu -x 2>&1 | tail -15 | cut -c1-300
timeout -k 5 1500 python -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/gpu_tests.log
tail -6 $O/gpu_tests.log | cut -c1-300
for rep in 1 2; do
  BADSLAM_HAND_OVER=1 python scripts/drop_in_profile.py 2>&1 | grep "ms per call" | sed 's/^/hand-over on:  /'
  BADSLAM_HAND_OVER=0 python scripts/drop_in_profile.py 2>&1 | grep "ms per call" | sed 's/^/hand-over off: /'
done
