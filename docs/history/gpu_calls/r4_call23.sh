#!/bin/bash
# round 4, call 23: shapes of the sorted record reduction of the intrinsics step (records per chunk / threads per workgroup):
# 4096/512 (default), 2048/256 (A), 4096/1024 (B), 2048/512 (C) -- stage time, then parity of each build
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_call23; mkdir -p $O
for round in 1 2; do
for v in default sortA sortB sortC; do
  if [ $v = default ]; then unset BADSLAM_LIB_DIR; else export BADSLAM_LIB_DIR=$GRAFT_REPO_ROOT/badslam_amd/lib_variants/$v; fi
  timeout -k 5 100 python bench.py --intrinsics --steps 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), d['stage_ms_per_iteration'])" | tee -a $O/timing.log
done
done
for v in sortA sortB sortC; do
  export BADSLAM_LIB_DIR=$GRAFT_REPO_ROOT/badslam_amd/lib_variants/$v
  echo "== $v" | tee -a $O/parity.log
  timeout -k 5 150 python -m pytest tests/test_gpu_intrinsics_pcg_vs_oracle.py tests/test_gpu_scale_parity.py -q -m gpu -x -k "intrinsics" 2>&1 | tail -2 | tee -a $O/parity.log
done
