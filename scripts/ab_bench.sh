#!/bin/bash
# scripts/ab_bench.sh <repeats> <variant dir or "-" for the in-tree build> ... : alternates bench.py runs of several builds on ONE
# GPU box (run-to-run and box-to-box noise is a few percent, more than most single optimisations) and prints it/s per run.
# Variants are build directories holding libbadslam_hip.so + libbadslam_host.so (badslam_amd/lib_variants/<name>, git-ignored).
# A variant written name:fast (or -:fast) runs with the fast arithmetic flavour (BAHIP_ARITHMETIC=fast; bench.py: BENCH_ARITHMETIC).
REPEATS=$1; shift
for r in $(seq $REPEATS); do
  for spec in "$@"; do
    v=${spec%%:*}; unset BAHIP_ARITHMETIC BENCH_ARITHMETIC; [ "$spec" != "$v" ] && export BAHIP_ARITHMETIC=${spec#*:} BENCH_ARITHMETIC=${spec#*:}
    if [ "$v" = "-" ]; then unset BADSLAM_LIB_DIR; else export BADSLAM_LIB_DIR=$PWD/badslam_amd/lib_variants/$v; fi
    python bench.py --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
s=d['stage_ms_per_iteration']
print('%-12s %7.1f it/s  %.3f ms/iter  geom %.3f  pose %.3f  solve %.3f  pose-launch %.3f' % ('$spec', d['value'], d['ms_per_step'], s['geometry_optimization'], s['pose_accumulate'], s['pose_solve'], d.get('roofline',{}).get('avg_launch_ms',0)))"
  done
done
