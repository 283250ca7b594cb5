#!/bin/bash
# round 4, call 22: kernel trace of the bench with its extras: where do the 44 ms of lifecycle per drop_in call go now?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r4_call22; mkdir -p $O
cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
ls -la $O/prof_bench/*/ | head
