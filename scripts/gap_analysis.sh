#!/bin/bash
# Kernel timeline of the timed BA iterations: where is the GPU idle?  usage: gap_analysis.sh [bench args]
REPO=$(pwd); OUT=$REPO/gpurun_out/gaps; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/t -- python $REPO/bench.py --no-cpu-baseline --steps 10 --warmup 2 "$@" > /dev/null 2> $OUT/log.txt
cd $REPO
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob('gpurun_out/gaps/t/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '').replace('bahip::', '')[:40]))
for f in glob.glob('gpurun_out/gaps/t/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'memcpy_' + r.get('Direction', '?')))
rows.sort()
# last 6 BA iterations: find activation kernels
act = [i for i, r in enumerate(rows) if r[2].startswith('activation_kernel')]
lo, hi = act[-7], act[-1]
seg = rows[lo:hi]
busy = sum(e - s for s, e, _ in seg)
span = seg[-1][1] - seg[0][0] + 0
print("6 iterations: span %.3f ms, busy %.3f ms, idle %.3f ms" % (span / 1e6, busy / 1e6, (span - busy) / 1e6))
prev_end = None
one = rows[act[-2]:act[-1]]
for s, e, n in one:
    gap = (s - prev_end) / 1e3 if prev_end else 0
    print("%8.1f us gap | %8.1f us %s" % (gap, (e - s) / 1e3, n))
    prev_end = max(prev_end or 0, e)
PY
