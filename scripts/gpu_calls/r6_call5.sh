#!/bin/bash
# round 6, call 5: is the 1/8 share bandwidth-bound by the REPLICATED image reads?  Contiguous and coarse-chunk surfel partitions (a rank's
# surfels then project into a part of the images only) against the chunk-cyclic partition of 4096 surfels
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r6_call5; mkdir -p $O
run() {  # chunk rank
  timeout -k 5 120 python bench.py --emulate-world 8 --emulate-rank $2 --shard-chunk $1 --force-allreduce --no-extras --no-cpu-baseline 2>/dev/null > $O/c$1_r$2.json
  python -c "
import json
d=json.load(open('$O/c$1_r$2.json')); s=d['stage_ms_per_iteration']; print('chunk %7d rank %d  %.4f ms  geom %.4f pose %.4f solve %.4f' % ($1, $2, d['ms_per_step'], s['geometry_optimization'], s['pose_accumulate'], s['pose_solve']))" | tee -a $O/partitions.txt
}
for c in 4096 16384 65536; do run $c 0; done
for r in 0 1 2 3 4 5 6 7; do run 0 $r; done
for r in 3 6; do run 65536 $r; done
