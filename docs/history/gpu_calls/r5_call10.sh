#!/bin/bash
# round 5, call 10: BASELINE configs[4] on one GPU (1280x960, 1000 keyframes, 20 M surfels, joint BA with the intrinsics step): the bench
# line, then rocprofv3 kernel stats + HBM counters of the same command
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call10; mkdir -p $O
C4="--width 1280 --height 960 --keyframes 1000 --surfels 20000000 --intrinsics --no-cpu-baseline --no-extras --steps 5 --warmup 2"
timeout -k 5 300 python bench.py $C4 > $O/config4.json 2> $O/config4.log
grep -v BindScene $O/config4.log | tail -5 | cut -c1-300; head -c 400 $O/config4.json; echo
PMC_STEPS=5 PASSES=basic PASS_TIMEOUT=280 bash scripts/profile_round.sh r5_config4 --width 1280 --height 960 --keyframes 1000 --surfels 20000000 --intrinsics --steps 5 --warmup 2 > $O/profile.log 2>&1
grep -A4 "timed region of the profile" $O/profile.log | cut -c1-500
head -14 $O/profile.log | cut -c1-160
