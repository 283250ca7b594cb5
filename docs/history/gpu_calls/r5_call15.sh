#!/bin/bash
# round 5, call 15: the device-driven loop is the default again (its switch's initialiser had been miscompiled since the split of capi.hip):
# routing tests, then device loop against host loop on one box, alternating
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call15; mkdir -p $O
timeout -k 5 600 python -m pytest tests/test_gpu_device_loop.py tests/test_gpu_sharded_loopback.py -q -m gpu -x 2>&1 | tail -5 | tee $O/gpu_tests.log | cut -c1-300
for r in 1 2 3; do
  for v in 1 0; do
    BAHIP_DEVICE_LOOP=$v timeout -k 5 200 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
s=d['stage_ms_per_iteration']
print('device_loop %s %7.1f it/s  %.3f ms/iter  geom %.3f  pose %.3f  solve %.3f  pose-launch %.3f  loop %s' % ('$v', d['value'], d['ms_per_step'], s['geometry_optimization'], s['pose_accumulate'], s['pose_solve'], d.get('roofline',{}).get('avg_launch_ms',0), (d['loop']['timed_calls_driven_by_the_device'], d['loop']['timed_calls_driven_by_the_host'])))" | tee -a $O/ab.txt
  done
done
