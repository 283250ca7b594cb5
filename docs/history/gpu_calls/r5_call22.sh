#!/bin/bash
# round 5, call 22: the run-order rebuild test, smoke()
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call22; mkdir -p $O
timeout -k 5 300 python -m pytest tests/test_gpu_tile_schedule.py -q -m gpu -x 2>&1 | tail -5 | tee $O/gpu_tests.log | cut -c1-300
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
