#!/bin/bash
# round 4, call 30: configs[4] again, with the record buffers' (re)allocations and long stream waits of the intrinsics step reported
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_HOST_TIMING=1
O=$GRAFT_REPO_ROOT/gpurun_out/r4_call30; mkdir -p $O
timeout -k 5 100 python bench.py --width 1280 --height 960 --keyframes 1000 --surfels 20000000 --intrinsics --no-cpu-baseline --no-extras > $O/config4.json 2> $O/config4.log
grep -v BindScene $O/config4.log | tail -30 | cut -c1-300; head -c 300 $O/config4.json
