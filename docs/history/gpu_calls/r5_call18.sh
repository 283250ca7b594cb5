#!/bin/bash
# round 5, call 18: is the timed region slower than its replay because it is the first sustained load of the process (clocks, first touch)?
# BENCH_PREPASS=1 runs warm-up + steps once before (discarded), resets the scene, then warms up and times as usual.  Alternating.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call18; mkdir -p $O
for r in 1 2 3; do
  for v in 0 1; do
    BENCH_PREPASS=$v timeout -k 5 200 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
s=d['stage_ms_per_iteration']
print('prepass %s %7.1f it/s  %.3f ms/iter | instrumented repeat %.3f ms/iter | geom %.3f  pose %.3f  solve %.3f  pose-launch %.3f' % ('$v', d['value'], d['ms_per_step'], d['instrumented_region']['ms_per_step'], s['geometry_optimization'], s['pose_accumulate'], s['pose_solve'], d.get('roofline',{}).get('avg_launch_ms',0)))" | tee -a $O/ab.txt
  done
done
