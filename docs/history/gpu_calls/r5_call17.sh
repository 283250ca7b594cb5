#!/bin/bash
# round 5, call 17: kernel trace of the PCG leg (where an outer iteration's time outside the step-1 sweep goes)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=8
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call17; mkdir -p $O
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $GRAFT_REPO_ROOT/bench.py --pcg --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/pcg.json 2> $O/pcg.err
tail -c 600 $O/pcg.json; echo
find $O/trace -name '*kernel_trace.csv' | head -3
f=$(find $O/trace -name '*kernel_trace.csv' | xargs ls -S | head -1)
cp $f $O/pcg_kernel_trace.csv
find $O/trace -type f -not -name '*kernel_trace.csv' -delete
ls -la $O/pcg_kernel_trace.csv
rm -rf $O/trace
