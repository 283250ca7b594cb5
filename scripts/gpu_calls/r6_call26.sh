#!/bin/bash
# round 6, call 26: the intrinsics sweep in 16 slices: parity, then configs[4] with 8 (automatic) and 16 slices (record memory against time)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r6_call26; mkdir -p $O
timeout -k 5 600 python -m pytest tests/test_gpu_intrinsics_pcg_vs_oracle.py -q -m gpu -x -k slices 2>&1 | tail -3
for S in 0 16; do
  BAHIP_INTR_SLICES=$S BADSLAM_HOST_TIMING=1 timeout -k 5 900 python bench.py --width 1280 --height 960 --keyframes 1000 --surfels 20000000 --intrinsics --no-cpu-baseline --no-extras > $O/config4_slices$S.json 2> $O/config4_slices$S.log
  grep 'intrinsics record buffers' $O/config4_slices$S.log | tail -2
  python -c "import json; d=json.load(open('$O/config4_slices$S.json')); print('slices $S:', round(d['value'],2), 'it/s', round(d['ms_per_step'],2), 'ms', d['stage_ms_per_iteration'])"
done
