#!/bin/bash
# Profiles `python bench.py` on the GPU box (run through gpurun) and leaves the summaries under
# gpurun_out/prof_<tag>/ ; copy what should be judged into profiles/.
#   scripts/profile_round.sh <tag> [bench args...]
# Pass 1: rocprofv3 --kernel-trace --stats (per-kernel time).  Passes 2,3: PMC counters alone
# (FETCH_SIZE and WRITE_SIZE need separate passes: TCC slots), no trace domains besides kernel-trace.
set -u
TAG=${1:-r1}
shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="python $REPO/bench.py --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH > "$OUT/bench_stats.json" 2> "$OUT/bench_stats.log"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > /dev/null 2> "$OUT/pmc_fetch.log"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH > /dev/null 2> "$OUT/pmc_write.log"
cd "$REPO"
python scripts/summarize_profile.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
# keep only the small files (kernel trace CSVs are large)
find "$OUT" -name '*kernel_trace.csv' -size +2M -delete
find "$OUT" -name '*counter_collection.csv' -size +2M -delete
