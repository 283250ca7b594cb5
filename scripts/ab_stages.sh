#!/bin/bash
# scripts/ab_stages.sh <repeats> <variant dir or "-"> ... : like ab_bench.sh, but prints every stage of stage_ms_per_iteration
# (BENCH_ARGS="--intrinsics ..." adds the intrinsics stage).
REPEATS=$1; shift
for r in $(seq $REPEATS); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset BADSLAM_LIB_DIR; else export BADSLAM_LIB_DIR=$PWD/badslam_amd/lib_variants/$v; fi
    python bench.py --no-cpu-baseline --no-extras $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
s=d['stage_ms_per_iteration']
print('%-14s %7.1f it/s  %.3f ms/iter | ' % ('$v', d['value'], d['ms_per_step']) + '  '.join('%s %.3f' % (k, v) for k, v in s.items()))"
  done
done
