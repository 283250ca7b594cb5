#!/bin/bash
# round 5, call 4: GPU suite with the final trims + strip-ordered planes; A/B against the round-4 HEAD build on one box; the new bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call4; mkdir -p $O
timeout -k 5 300 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $O/gpu_tests.log
tail -5 $O/gpu_tests.log
BENCH_ARGS="--no-extras" timeout -k 5 300 bash scripts/ab_bench.sh 3 r4head - 2>&1 | tee $O/ab.txt
timeout -k 5 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5_call4/bench.json"))
print(d["value"], d.get("events_off"), d.get("launch_window"))
print({k: d["roofline"][k] for k in ("frac","avg_launch_ms","launches","keyframes_per_launch")})
print(d.get("drop_in",{}).get("ms_per_call"), d.get("cold_start",{}).get("ba_iterations_per_s"), d.get("pcg",{}).get("outer_iterations_per_s"), d.get("intrinsics"))
PY
tail -3 $O/bench.err
