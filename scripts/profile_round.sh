#!/bin/bash
# Profiles `python bench.py` on the GPU box (run through gpurun) and leaves the summaries under gpurun_out/prof_<tag>/ ;
# copy what should be judged into profiles/ (scripts/summarize_profile.py prints the cp line).
#   scripts/profile_round.sh <tag> [bench args...]
# Pass 1: rocprofv3 --kernel-trace --stats (per-kernel time; the bench line of the same command is kept).
# Passes 2-4: PMC counters alone, one pass each (FETCH_SIZE and WRITE_SIZE share TCC slots), --kernel-trace only as the pool
# requires: FETCH_SIZE, WRITE_SIZE, and the issue-side counters behind "VALU-issue bound".
# Passes 5-6: the VALU instruction mix (binary32 add / mul / fma / transcendental; integer, conversion and binary64 counts), from
# which summarize_profile.py derives binary32 flops per launch and the share of VALU instructions that are not arithmetic.
# Pass 7: FETCH_SIZE over reads of known size (scripts/fetch_calibration.py): what the counter reports per byte at the
# 4-byte access width of the sweeps.
set -u
TAG=${1:-r2}
shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export BADSLAM_RENDER_WORKERS=${BADSLAM_RENDER_WORKERS:-8}   # rocprofv3 attaches to every spawned render worker; a run with 96 of them hung once
PASSES=${PASSES:-all}                                         # all | basic (stats, FETCH_SIZE, WRITE_SIZE, calibration) | stats
[ -n "${ONLY_STATS:-}" ] && PASSES=stats
LIMIT="timeout -k 5 ${PASS_TIMEOUT:-150}"                     # a pass that hangs must not take the GPU budget with it (-k: rocprofv3 traps SIGTERM)
cd /tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --no-extras $*"
$LIMIT rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH > "$OUT/bench_stats.json" 2> "$OUT/bench_stats.log"
if [ "$PASSES" != stats ]; then   # PASSES=stats (or ONLY_STATS=1): the kernel-time table alone
$LIMIT rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH --steps ${PMC_STEPS:-10} > "$OUT/bench_pmc_fetch.json" 2> "$OUT/pmc_fetch.log"
$LIMIT rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH --steps ${PMC_STEPS:-10} > "$OUT/bench_pmc_write.json" 2> "$OUT/pmc_write.log"
[ "$PASSES" = all ] && $LIMIT rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d "$OUT/pmc_sq" -- $BENCH --steps ${PMC_STEPS:-10} > "$OUT/bench_pmc_sq.json" 2> "$OUT/pmc_sq.log"
[ "$PASSES" = all ] && $LIMIT rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d "$OUT/pmc_flops" -- $BENCH --steps ${PMC_STEPS:-10} > "$OUT/bench_pmc_flops.json" 2> "$OUT/pmc_flops.log"
[ "$PASSES" = all ] && $LIMIT rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 --output-format csv -d "$OUT/pmc_mix" -- $BENCH --steps ${PMC_STEPS:-10} > "$OUT/bench_pmc_mix.json" 2> "$OUT/pmc_mix.log"
if [ -n "${CAL_FROM:-}" ] && [ -d "$CAL_FROM/pmc_cal" ]; then   # the calibration of another profile of the same call
  cp -r "$CAL_FROM/pmc_cal" "$OUT/pmc_cal"; cp "$CAL_FROM/cal_bytes.txt" "$OUT/cal_bytes.txt"
else
$LIMIT rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_cal" -- python $REPO/scripts/fetch_calibration.py > "$OUT/cal_bytes.txt" 2> "$OUT/pmc_cal.log"
fi
fi
cd "$REPO"
python scripts/summarize_profile.py "$OUT" "$TAG" $* > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
# (every pass keeps its own bench line: summarize_profile.py reads the pass's "launch_window" to sum the counters over the
# dispatches of that run's timed region and divides by that run's own launches / keyframes visited / iterations)
# keep only the small files (kernel trace CSVs are large)
find "$OUT" -name '*kernel_trace.csv' -size +2M -delete
find "$OUT" -name '*counter_collection.csv' -size +2M -delete
