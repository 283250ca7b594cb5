#!/bin/bash
# round 4, call 1: LDS sink shapes (root cause of the r3 anomaly), queued-ahead rounds, LDS parts; then the whole GPU suite; A/B bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4_call1; mkdir -p $O
echo "== in-tree: LDS matrix"; timeout 900 python -m pytest tests/test_gpu_scale_parity.py -x -q -m gpu -k "batched_pose_estimation" 2>&1 | tail -5 | tee $O/lds_matrix_intree.log
for v in asm_imm builtin_def; do
  echo "== variant $v: LDS matrix"
  BADSLAM_LIB_DIR=$PWD/badslam_amd/lib_variants/$v timeout 900 python -m pytest tests/test_gpu_scale_parity.py -q -m gpu -k "batched_pose_estimation and lds" 2>&1 | tail -8 | tee $O/lds_matrix_$v.log
  BADSLAM_LIB_DIR=$PWD/badslam_amd/lib_variants/$v timeout 600 python -m pytest tests/test_gpu_kernels_vs_oracle.py -q -m gpu -k "pose" 2>&1 | tail -4 | tee -a $O/lds_matrix_$v.log
done
echo "== full gpu suite"; timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $O/gpu_tests.log
echo "== A/B bench"; BENCH_ARGS="--no-extras" timeout 900 bash scripts/ab_bench.sh 2 - r3_shape builtin_def 2>&1 | tee $O/ab_bench.log
echo "== emulate world 8"
for v in - r3_shape; do
  if [ "$v" = "-" ]; then unset BADSLAM_LIB_DIR; else export BADSLAM_LIB_DIR=$PWD/badslam_amd/lib_variants/$v; fi
  for w in 8 4; do
  timeout 300 python bench.py --no-cpu-baseline --no-extras --emulate-world $w --force-allreduce 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms_per_iteration']
print('$v world $w: %.3f ms/iter | ' % d['ms_per_step'] + '  '.join('%s %.3f' % (k, v) for k, v in s.items()), d['roofline']['launches_by_form'])" | tee -a $O/emulate.log
  done
done
