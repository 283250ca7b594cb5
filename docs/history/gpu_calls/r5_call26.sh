#!/bin/bash
# round 5, call 26: 24 of the intrinsics sweep's 34 per-lane sums in LDS (127 VGPRs, 4 wavefronts per SIMD) against all in registers
# (151 VGPRs, 3 wavefronts: variant intr0): parity, then the stage at the bench size, alternating
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call26; mkdir -p $O
timeout -k 5 400 python -m pytest tests/test_gpu_intrinsics_pcg_vs_oracle.py tests/test_gpu_scale_parity.py tests/test_gpu_sharded_loopback.py -q -m gpu -x -k "intrinsics" 2>&1 | tail -4 | tee $O/gpu_tests.log | cut -c1-300
for r in 1 2 3; do
  for v in intr0 -; do
    if [ "$v" = "-" ]; then unset BADSLAM_LIB_DIR; else export BADSLAM_LIB_DIR=$PWD/badslam_amd/lib_variants/$v; fi
    timeout -k 5 200 python bench.py --no-cpu-baseline --no-extras --intrinsics --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
s=d['stage_ms_per_iteration']
print('%-6s value %.1f  %.4f ms/iter  intrinsics stage %.4f ms  (geom %.3f pose %.3f)' % ('$v', d['value'], d['ms_per_step'], s['intrinsics_optimization'], s['geometry_optimization'], s['pose_accumulate']))" | tee -a $O/ab.txt
  done
done
unset BADSLAM_LIB_DIR
