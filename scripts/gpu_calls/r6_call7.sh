#!/bin/bash
# round 6, call 7: two wavefronts per tile in the fused geometry launch on the emulated shares
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r6_call7; mkdir -p $O
run() {  # label world env...
  label=$1; w=$2; shift 2
  env "$@" timeout -k 5 120 python bench.py --emulate-world $w --force-allreduce --no-extras --no-cpu-baseline 2>/dev/null > $O/${label}_w$w.json
  python -c "
import json
d=json.load(open('$O/${label}_w$w.json')); s=d['stage_ms_per_iteration']; print('%-22s world %d  %.4f ms  geom %.4f pose %.4f solve %.4f' % ('$label', $w, d['ms_per_step'], s['geometry_optimization'], s['pose_accumulate'], s['pose_solve']))" | tee -a $O/waves.txt
}
for w in 8 4 2; do
  run waves4 $w BAHIP_GEOMETRY_SPLIT=0 BAHIP_TILE_WAVES=4
  run waves2 $w BAHIP_GEOMETRY_SPLIT=0 BAHIP_TILE_WAVES=2
  run waves1 $w BAHIP_GEOMETRY_SPLIT=0 BAHIP_TILE_WAVES=1
done
