#!/bin/bash
# round 5, call 12: the intrinsics step in slices (records of a slice reduced on a second stream while the next slice sweeps): parity tests,
# the stage at the bench size with 1 / 2 / 5 slices, and BASELINE configs[4]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call12; mkdir -p $O
timeout -k 5 300 python -m pytest tests/test_gpu_intrinsics_pcg_vs_oracle.py tests/test_gpu_scale_parity.py -q -m gpu -x -k "intrinsics" 2>&1 | tail -6 | tee $O/gpu_tests.log | cut -c1-300
for s in 1 2 5 0; do
  BAHIP_INTR_SLICES=$s timeout -k 5 200 python bench.py --no-cpu-baseline --no-extras --intrinsics --steps 10 > $O/intr_$s.json 2> $O/intr_$s.err
  python - $s <<'PY'
import json, sys
d=json.load(open("gpurun_out/r5_call12/intr_%s.json" % sys.argv[1]))
print("slices", sys.argv[1], "value", round(d["value"],1), "ms/iter", round(d["ms_per_step"],4), {k: round(v,4) for k,v in d["stage_ms_per_iteration"].items()})
PY
done
C4="--width 1280 --height 960 --keyframes 1000 --surfels 20000000 --intrinsics --no-cpu-baseline --no-extras --steps 5 --warmup 2"
BADSLAM_HOST_TIMING=1 timeout -k 5 300 python bench.py $C4 > $O/config4.json 2> $O/config4.log
grep "record buffers" $O/config4.log | tail -2 | cut -c1-200
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5_call12/config4.json"))
print("config4", round(d["value"],2), "it/s", round(d["ms_per_step"],2), "ms", {k: round(v,3) for k,v in d["stage_ms_per_iteration"].items()})
PY
