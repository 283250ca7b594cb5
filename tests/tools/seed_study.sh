#!/bin/bash
# tests/tools/seed_study.sh -- VERDICT r1 "weak 4": two of the reference's closed-loop tests pass for some random scenes and miss
# their bound for others.  Is the miss convergence speed or a biased optimum?  Runs both tests for seeds 1..7 with the
# prescribed number of BundleAdjustment calls and with 4x as many, with exact binary32 bilinear weights (the product build)
# and, for the photometric test, with weights rounded to 8 fractional bits like CUDA's tex2D (the experiment build under
# badslam_amd/lib_variants/quantized, made with -DBAHIP_QUANTIZED_BILINEAR_WEIGHTS).  Prints the final estimates.
cd "$(dirname "$0")/../.."
for variant in lib lib_variants/quantized; do
  BIN=badslam_amd/$variant/test_directba
  [ -x "$BIN" ] || continue
  for factor in 1 4; do
    for seed in 1 2 3 4 5 6 7; do
      line=$(TEST_SEED=$seed TEST_CALLS_FACTOR=$factor $BIN AlternatingIntrinsicsOptimizationWithPhotometricResidual 2>&1 | grep camera_difference | tail -1)
      echo "photometric-intrinsics $variant calls_x$factor seed $seed: $line"
    done
  done
done
for factor in 1 4; do
  for seed in 1 2 3 4 5 6 7; do
    line=$(TEST_SEED=$seed TEST_CALLS_FACTOR=$factor badslam_amd/lib/test_directba AlternatingDepthDeformationOptimizationWithGeometricResidual 2>&1 | grep "call " | tail -1)
    echo "depth-deformation calls_x$factor seed $seed: $line"
  done
done
