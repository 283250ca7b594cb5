#!/bin/bash
# round 4, call 28: BASELINE configs[4] on one GPU (1280x960, 1000 keyframes, 20 M surfels, joint BA with the intrinsics step)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_call28; mkdir -p $O
timeout -k 5 170 python bench.py --width 1280 --height 960 --keyframes 1000 --surfels 20000000 --intrinsics --no-cpu-baseline --no-extras --steps 5 --warmup 2 > $O/config4.json 2> $O/config4.log
tail -3 $O/config4.log; head -c 900 $O/config4.json
