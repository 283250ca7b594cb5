#!/bin/bash
# round 4, call 27: compile-flag variants of the whole backend (scheduler strategy / bias / trackers / -O2) against the default build,
# alternating runs of the default bench on one box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_call27; mkdir -p $O
for round in 1 2; do
for v in default maxilp bias0 trackers o2; do
  if [ $v = default ]; then unset BADSLAM_LIB_DIR; else export BADSLAM_LIB_DIR=$GRAFT_REPO_ROOT/badslam_amd/lib_variants/$v; fi
  timeout -k 5 100 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],4), {k:round(x,4) for k,x in d['stage_ms_per_iteration'].items()})" | tee -a $O/timing.log
done
done
