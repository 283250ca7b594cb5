#!/bin/bash
# round 5, call 34: the association test of the sweeps as ANDed verdicts instead of a chain of early returns (no save-exec / branch per
# comparison): parity of every sweep, then A/B against the build before (base), alternating
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call34; mkdir -p $O
timeout -k 5 600 python -m pytest tests/test_gpu_kernels_vs_oracle.py tests/test_gpu_scale_parity.py tests/test_gpu_sharded_loopback.py tests/test_gpu_golden_reference.py tests/test_gpu_intrinsics_pcg_vs_oracle.py tests/test_gpu_e2e_vga.py tests/test_gpu_lifecycle_stages.py tests/test_gpu_directba_vs_oracle.py -q -m gpu -x 2>&1 | tail -4 | tee $O/gpu_tests.log | cut -c1-300
BENCH_ARGS="--no-extras" bash scripts/ab_bench.sh 3 base - 2>&1 | tee $O/ab.txt
