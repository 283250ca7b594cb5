#!/bin/bash
# round 6, call 23: BASELINE configs[4] on one GPU with the round's code (1280x960, 1000 keyframes, 20 M surfels, joint BA with the intrinsics step)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r6_call23; mkdir -p $O
timeout -k 5 1500 python bench.py --width 1280 --height 960 --keyframes 1000 --surfels 20000000 --intrinsics --no-cpu-baseline --no-extras > $O/config4_line.json 2> $O/config4.log
tail -5 $O/config4.log | cut -c1-300
python - <<PY
import json
d=json.load(open("$O/config4_line.json"))
print(round(d["value"],2), "it/s", round(d["ms_per_step"],2), "ms;", d["config"].get("arithmetic"))
print("roofline", {k:d["roofline"].get(k) for k in ("kernel","avg_launch_ms","frac")})
print("roofline_intrinsics", d.get("roofline_intrinsics"))
print("stage", d.get("stage_ms_per_iteration"))
PY
