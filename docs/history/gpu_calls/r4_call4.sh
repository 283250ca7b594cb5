#!/bin/bash
# round 4, call 4: the bisect proper -- the failing commit with its range limit fixed (v4: fails at K = 200, rows > 0) plus the pin (v5),
# plus the batch word moved behind the table (v6), plus both (v7)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_call4; mkdir -p $O
for v in v4 v5 v6 v7; do
  echo "== bisect $v"
  (cd _bisect/$v && timeout 600 python -m pytest tests/test_gpu_kernels_vs_oracle.py tests/test_gpu_scale_parity.py -q -m gpu -k "lds" 2>&1 | tail -4) | tee $O/bisect_$v.log
done
timeout 600 python -m pytest tests/test_gpu_device_loop.py -q -m gpu 2>&1 | tail -3
