#!/bin/bash
# round 4, call 29: BASELINE configs[4] on one GPU with the bench's own step counts (20 timed iterations after 3 warm-up: comparable
# with profiles/r3_c_config4_bench.json; call 28 timed iterations 3-7 of a cold start plus the end tasks of its one call)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_HOST_TIMING=1
O=$GRAFT_REPO_ROOT/gpurun_out/r4_call29; mkdir -p $O
timeout -k 5 150 python bench.py --width 1280 --height 960 --keyframes 1000 --surfels 20000000 --intrinsics --no-cpu-baseline --no-extras > $O/config4.json 2> $O/config4.log
grep -v BindScene $O/config4.log | tail -8 | cut -c1-400; head -c 500 $O/config4.json
