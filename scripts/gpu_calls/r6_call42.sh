#!/bin/bash
# round 6, call 42: rounds queued per pose phase by the device-driven loop: automatic against 2 and 3 forced; how long the hand-overs take
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r6_call42; mkdir -p $O
BADSLAM_HOST_TIMING=1 timeout -k 5 300 python bench.py --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep 'bahip_alternating_iterations, us' | tail -6
for rep in 1 2; do
  for R in 0 2 3; do
    BAHIP_POSE_ROUNDS_AHEAD=$R timeout -k 5 300 python bench.py --no-cpu-baseline --no-extras > $O/w1_R${R}_$rep.json 2> /dev/null
    BAHIP_POSE_ROUNDS_AHEAD=$R timeout -k 5 300 python bench.py --emulate-world 8 --force-allreduce --no-cpu-baseline --no-extras > $O/w8_R${R}_$rep.json 2> /dev/null
    python -c "import json; a=json.load(open('$O/w1_R${R}_$rep.json')); b=json.load(open('$O/w8_R${R}_$rep.json')); print('rounds ahead $R rep $rep: one GPU', round(a['value'],1), 'it/s; world 8 share', round(b['ms_per_step'],4), 'ms; pose dispatches', a['launch_window'].get('pose_dispatches_timed'), b['launch_window'].get('pose_dispatches_timed'))"
  done
done
