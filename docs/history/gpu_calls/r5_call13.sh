#!/bin/bash
# round 5, call 13: the tile census of the first pose round in shader cycles (BAHIP_TILE_COST_SHIFT) against candidates visited,
# alternating runs on one box; then the full GPU suite on the same build
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call13; mkdir -p $O
for r in 1 2 3; do
  for v in none 11 13; do
    if [ "$v" = "none" ]; then unset BAHIP_TILE_COST_SHIFT; else export BAHIP_TILE_COST_SHIFT=$v; fi
    timeout -k 5 200 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
s=d['stage_ms_per_iteration']
print('shift %-5s %7.1f it/s  %.3f ms/iter  geom %.3f  pose %.3f  solve %.3f  pose-launch %.3f' % ('$v', d['value'], d['ms_per_step'], s['geometry_optimization'], s['pose_accumulate'], s['pose_solve'], d.get('roofline',{}).get('avg_launch_ms',0)))" | tee -a $O/ab.txt
  done
done
unset BAHIP_TILE_COST_SHIFT
timeout -k 5 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -4 | tee $O/gpu_tests.log | cut -c1-300
