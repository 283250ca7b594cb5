#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_call12; mkdir -p $O
(cd _bisect/pC && timeout 120 python diag_adds.py 2>&1 | tail -12) | tee $O/diag_adds.log
