#!/bin/bash
# round 6, call 1: A/B of the fast-math variant (raw v_rcp / v_sqrt / v_exp, contraction, FTZ) against the exact build on one box;
# the reference-kernel goldens and the e2e VGA test under the fast library
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r6_call1; mkdir -p $O
BENCH_ARGS="--no-extras" timeout -k 5 400 bash scripts/ab_bench.sh 3 - fast 2>&1 | tee $O/ab.txt
BADSLAM_LIB_DIR=$PWD/badslam_amd/lib_variants/fast timeout -k 5 600 python -m pytest tests/test_gpu_golden_reference.py tests/test_gpu_e2e_vga.py tests/test_gpu_directba_cpp.py tests/test_gpu_tum_pipeline.py -q -m gpu 2>&1 | tail -40 > $O/fast_tests.log
tail -30 $O/fast_tests.log | cut -c1-400
