#!/bin/bash
# round 4, call 18: kernel timeline of the emulated 8-rank share with the rounds policy in place (what fills 0.41 ms?)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=8
O=$GRAFT_REPO_ROOT/gpurun_out/r4_call18; mkdir -p $O
cd /tmp
timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_emu8 -- python $GRAFT_REPO_ROOT/bench.py --emulate-world 8 --force-allreduce --no-cpu-baseline --no-extras > $O/emu8_prof.json 2> $O/emu8_prof.err
cd $GRAFT_REPO_ROOT
head -c 400 $O/emu8_prof.json
