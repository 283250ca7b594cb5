#!/bin/bash
# round 4, call 26: full GPU suite and smoke on the round's last commit
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_call26; mkdir -p $O
timeout -k 5 400 python -m pytest tests -q -m gpu 2>&1 | tail -25 > $O/gpu_tests.log
tail -3 $O/gpu_tests.log
timeout -k 5 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
