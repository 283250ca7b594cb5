#!/bin/bash
# round 6, call 6: the split form of the geometry step ((tile, class) workgroups, three launches) against the fused four-wavefront form
# on the emulated shares; pose units per tile on the shares; parity of the split form
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r6_call6; mkdir -p $O
timeout -k 5 900 python -m pytest tests/test_gpu_scale_parity.py tests/test_gpu_fast_flavour.py tests/test_gpu_sharded_loopback.py tests/test_gpu_kernels_vs_oracle.py tests/test_gpu_device_loop.py -q -m gpu -k "not c3 and not c2_size and not c5" 2>&1 | tail -8
run() {  # label world env...
  label=$1; w=$2; shift 2
  env "$@" timeout -k 5 120 python bench.py --emulate-world $w --force-allreduce --no-extras --no-cpu-baseline 2>/dev/null > $O/${label}_w$w.json
  python -c "
import json
d=json.load(open('$O/${label}_w$w.json')); s=d['stage_ms_per_iteration']; print('%-22s world %d  %.4f ms  geom %.4f pose %.4f solve %.4f' % ('$label', $w, d['ms_per_step'], s['geometry_optimization'], s['pose_accumulate'], s['pose_solve']))" | tee -a $O/split.txt
}
for w in 8 4 2; do
  run fused $w BAHIP_GEOMETRY_SPLIT=0
  run split $w BAHIP_GEOMETRY_SPLIT=1
done
run split_parts1 8 BAHIP_GEOMETRY_SPLIT=1 BAHIP_POSE_LDS_PARTS_SHIFT=1
run split_parts0 8 BAHIP_GEOMETRY_SPLIT=1 BAHIP_POSE_LDS_PARTS_SHIFT=0
run split_parts3 8 BAHIP_GEOMETRY_SPLIT=1 BAHIP_POSE_LDS_PARTS_SHIFT=3
run split_fast 8 BAHIP_GEOMETRY_SPLIT=1 BENCH_ARITHMETIC=fast
run split_fast 4 BAHIP_GEOMETRY_SPLIT=1 BENCH_ARITHMETIC=fast
