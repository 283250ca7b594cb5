#!/bin/bash
# round 4, call 14: full GPU suite on the current tree; default bench line with the new extras; emulated 8-rank share; kernel stats of
# a bench run (lifecycle kernels of the drop_in leg included) and the host-side timing of that leg
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_call14; mkdir -p $O
timeout 600 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $O/gpu_tests.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 200 python bench.py --emulate-world 8 --force-allreduce --no-cpu-baseline --no-extras > $O/bench_emu8.json 2> $O/bench_emu8.err
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o dropin -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/prof -name "*.db" -delete; find $O/prof -name "*kernel_trace.csv" -delete
tail -3 $O/gpu_tests.log; cat $O/bench_default.json | head -c 6000; echo; cat $O/bench_emu8.json | head -c 1500
