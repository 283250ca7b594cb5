#!/bin/bash
# round 6, call 11: HEAD of the re-entered session: full GPU suite, the default bench line, the round's counter evidence (profile_all r6_a)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r6_call11; mkdir -p $O
timeout -k 5 1500 python -m pytest tests -q -m gpu --durations=10 2>&1 | tail -60 > $O/gpu_tests.log
tail -5 $O/gpu_tests.log | cut -c1-300
timeout -k 5 500 python bench.py > $O/bench.json 2> $O/bench.log
cut -c1-1500 $O/bench.json
BADSLAM_RENDER_WORKERS=8 timeout -k 5 1500 bash scripts/profile_all.sh r6_a > $O/profile_all.log 2>&1
tail -30 $O/profile_all.log | cut -c1-300
