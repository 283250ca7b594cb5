#!/bin/bash
# round 5, call 41: scene binding without host waits (page-locked upload stages): the full GPU suite, then A/B of the bench line and the drop-in figure
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call41; mkdir -p $O
timeout -k 5 700 python -m pytest tests -q -m gpu -x 2>&1 | tail -4 | tee $O/gpu_tests.log | cut -c1-300
BENCH_ARGS="--no-extras" bash scripts/ab_bench.sh 3 base - 2>&1 | tee $O/ab.txt
for v in base -; do
  if [ "$v" = "-" ]; then unset BADSLAM_LIB_DIR; else export BADSLAM_LIB_DIR=$PWD/badslam_amd/lib_variants/$v; fi
  echo "$v $(python scripts/drop_in_profile.py 2>&1 | grep 'ms per call')" | tee -a $O/ab.txt
done
