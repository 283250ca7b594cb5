#!/bin/bash
# round 4, call 10: parts of the commit from which the immediate LDS adds are right, each alone on the commit before it (pA: the batch
# protocol with a size field; pB: the per-tile visit counter); and the PCG step-1 sweep in its persistent LDS form
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_call10; mkdir -p $O
for v in pA pB; do
  echo "== $v"
  (cd _bisect/$v && timeout 600 python -m pytest tests/test_gpu_scale_parity.py -q -m gpu -k "batched_pose_estimation and lds" 2>&1 | tail -2) | tee -a $O/bisect_parts.log
done
echo "== PCG tests"; timeout 1200 python -m pytest tests -q -m gpu -k "pcg or PCG" 2>&1 | tail -6 | tee $O/pcg_tests.log
for f in 1 0; do
BAHIP_PCG_LDS=$f timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('pcg_lds=$f', d['value'], d.get('pcg'))" | tee -a $O/pcg_bench.log
done
