#!/bin/bash
# round 5, call 38: BASELINE configs[4] on one GPU with the round's last kernels (the line only; counters: profiles/r5_config4_* of call 10)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call38; mkdir -p $O
C4="--width 1280 --height 960 --keyframes 1000 --surfels 20000000 --intrinsics --no-cpu-baseline --no-extras --steps 5 --warmup 2"
timeout -k 5 400 python bench.py $C4 > $O/config4.json 2> $O/config4.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5_call38/config4.json"))
print("config4", round(d["value"],2), "it/s", round(d["ms_per_step"],2), "ms", {k: round(v,3) for k,v in d["stage_ms_per_iteration"].items()}, d["intrinsics"] if "intrinsics" in d else "")
PY
