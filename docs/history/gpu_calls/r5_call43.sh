#!/bin/bash
# round 5, call 43: the emulated 8-rank share again, four times (call 42 read 0.411 ms in the timed region against 0.370 in its instrumented repeat)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call43; mkdir -p $O
for r in 1 2 3 4; do
  timeout -k 5 100 python bench.py --emulate-world 8 --force-allreduce --no-cpu-baseline --no-extras > $O/bench_emu8_$r.json 2> $O/err_$r.txt
  python -c "
import json; d=json.load(open('$O/bench_emu8_$r.json')); print('emulated world 8 run $r:', round(d['ms_per_step'],4), 'timed,', round(d['instrumented_region']['ms_per_step'],4), 'instrumented repeat')"
done
