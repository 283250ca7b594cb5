#!/bin/bash
# round 5, call 23: the pose sweep reads a candidate's image pointers from frames[w] (no dependent load of work[w].kf_index): parity tests, A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call23; mkdir -p $O
timeout -k 5 600 python -m pytest tests/test_gpu_kernels_vs_oracle.py tests/test_gpu_scale_parity.py tests/test_gpu_device_loop.py tests/test_gpu_sharded_loopback.py -q -m gpu -x 2>&1 | tail -4 | tee $O/gpu_tests.log | cut -c1-300
BENCH_ARGS="--no-extras" bash scripts/ab_bench.sh 3 base - 2>&1 | tee $O/ab.txt
