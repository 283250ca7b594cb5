#!/bin/bash
# scripts/scale_curve.sh [ranks ...] -- the multi-GPU scaling table of the BA hot path on ONE node, both sharding axes, native RCCL
# (ncclAllReduce on the backend's own stream over xGMI), one process per GPU exactly as the driver launches bench.py:
#   python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
# Prints, per axis and rank count: BA iterations/s, ms per iteration, speed-up over the 1-GPU line of the same run, the exchange
# (calls and bytes per iteration) and the slowest rank's stage times; the full JSON lines go to $OUT (default gpurun_out/scale_curve).
# On a box with fewer devices than ranks it says so and, with EMULATE=1, prints rank 0's SHARE of an N-rank run instead (one GPU,
# exchange path through a one-rank communicator: compute share only, no link time) -- a planning aid, labelled as such.
# BASELINE.json configs[2] (1 GPU) and configs[3] (keyframes over 8 GPUs) are the default workload; STEPS / WARMUP / EXTRA pass through.
set -u
cd "$(dirname "$0")/.."
RANKS=${*:-1 2 4 8}
OUT=${OUT:-gpurun_out/scale_curve}; mkdir -p "$OUT"
STEPS=${STEPS:-20}; WARMUP=${WARMUP:-3}; EXTRA=${EXTRA:-}
DEVICES=$(python - <<'PY'
from badslam_amd import capi
print(int(capi.load().bahip_device_count()))
PY
)
echo "devices on this node: $DEVICES"
row() {  # axis ranks json-file
  python - "$1" "$2" "$3" "${BASE:-}" <<'PY'
import json, sys
axis, ranks, path, base = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
try:
    d = json.loads(open(path).read().strip().splitlines()[-1])
except Exception as e:
    print(f"{axis:10s} {ranks:2d}  no result ({e})"); sys.exit(0)
speedup = f"{d['value'] / float(base):5.2f}x" if base else "   -  "
share = " (rank 0's share, emulated)" if "emulated_share_of_world" in d else ""
ex = d.get("exchange")
ex_s = f"  exchange {ex['calls_per_iteration']:.1f} calls / {ex['bytes_per_iteration'] / 1e6:.2f} MB per iteration" if ex else ""
stages = d.get("stage_ms_per_iteration", {})
if d.get("per_rank"):
    slow = max(d["per_rank"], key=lambda r: r["ms_per_step_before_barrier"])
    stages = slow["stage_ms_per_iteration"]
st = "  ".join(f"{k} {v:.3f}" for k, v in stages.items() if v)
print(f"{axis:10s} {ranks:2d}  {d['value']:8.1f} it/s  {d['ms_per_step']:.3f} ms/iter  {speedup}{share}{ex_s}  | {st}")
if axis == "surfels" and ranks == 1:
    open(path + ".base", "w").write(str(d["value"]))
PY
}
for AXIS in surfels keyframes; do
  for N in $RANKS; do
    F="$OUT/${AXIS}_${N}.json"
    if [ "$N" = 1 ]; then
      [ "$AXIS" = keyframes ] && continue
      python bench.py --no-cpu-baseline --no-extras --steps $STEPS --warmup $WARMUP $EXTRA > "$F" 2> "$F.err"
    elif [ "$DEVICES" -ge "$N" ]; then
      PORT=$((29500 + RANDOM % 2000))
      python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus $N --shard $AXIS \
        --no-cpu-baseline --no-extras --steps $STEPS --warmup $WARMUP $EXTRA > "$F" 2> "$F.err"
    elif [ "${EMULATE:-0}" = 1 ]; then
      python bench.py --no-cpu-baseline --no-extras --emulate-world $N --force-allreduce --shard $AXIS --steps $STEPS --warmup $WARMUP $EXTRA > "$F" 2> "$F.err"
    else
      echo "$AXIS $N: this node has $DEVICES device(s); nothing measured (EMULATE=1 prints rank 0's share on one GPU instead)"; continue
    fi
    BASE=$(cat "$OUT/surfels_1.json.base" 2>/dev/null) row $AXIS $N "$F"
  done
done
