#!/bin/bash
# round 5, call 16: where the per-call time of the default bench goes on the host side (BADSLAM_HOST_TIMING prints)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call16; mkdir -p $O
BADSLAM_HOST_TIMING=1 timeout -k 5 200 python bench.py --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err
grep "us\]" $O/bench.err | tail -24 | cut -c1-220
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['stage_ms_per_iteration'], d['loop'])"
