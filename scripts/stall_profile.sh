#!/bin/bash
# Issue-side PMC passes for the two sweeps (run through gpurun): where do the wave-cycles go?  Output: gpurun_out/stall_<tag>/
#   scripts/stall_profile.sh <tag> [bench args]
set -u
TAG=${1:-x}; shift || true
REPO=$(pwd); OUT=$REPO/gpurun_out/stall_$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 1 $*"
# every pass under its own time limit: pass d has hung twice on the pool (r3_a, r3_c) and took the rest of the call's budget with it
pass() { n=$1; shift; timeout ${PASS_TIMEOUT:-60} rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$n" -- $BENCH > /dev/null 2> "$OUT/$n.log" || echo "pass $n: not completed" >&2; }
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pass b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH
pass c SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT
pass d SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU
pass e SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT
pass f SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64
cd "$REPO"
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:60]
        a = agg[k][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", [0, 1])[0])[:4]:
    print(k)
    for c in sorted(agg[k]):
        v, n = agg[k][c]
        print("   %-26s %.4g per launch (%d launches)" % (c, v / n, n))
PY
find "$OUT" -name '*.csv' -size +2M -delete
