#!/bin/bash
# round 5, call 14: which loop does the default bench take (device-driven or host)?  host-timing prints of a small run
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call14; mkdir -p $O
BADSLAM_HOST_TIMING=1 timeout -k 5 200 python bench.py --no-cpu-baseline --no-extras --keyframes 24 --surfels 300000 --steps 4 --warmup 2 > $O/small.json 2> $O/small.err
grep "DirectBA\]\|bahip_alternating" $O/small.err | sort | uniq -c | head -20
