#!/bin/bash
# round 6, call 22: hybrid parts in the pose sweep, heavy threshold of small clouds: parity, then the emulated 8-rank share per setting
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r6_call22; mkdir -p $O
timeout -k 5 900 python -m pytest tests/test_gpu_device_loop.py tests/test_gpu_kernels_vs_oracle.py tests/test_gpu_sharded_loopback.py -q -m gpu -x 2>&1 | tail -5 | cut -c1-300
run() {  # name, env...
  name=$1; shift
  env "$@" timeout -k 5 300 python bench.py --emulate-world 8 --force-allreduce --no-cpu-baseline --no-extras > $O/emu8_$name.json 2> $O/emu8_$name.log
  python - <<PY
import json
d=json.load(open("$O/emu8_$name.json"))
st=d.get("stage_ms_per_iteration",{})
print("world 8 $name:", round(d["ms_per_step"],4), "ms per iteration;", {k: round(v,4) for k,v in st.items() if isinstance(v,(int,float))})
PY
}
for rep in 1 2; do
  run base_$rep BAHIP_POSE_HYBRID_PARTS=0
  run posehyb_$rep BAHIP_POSE_HYBRID_PARTS=-1
  run posehyb3_$rep BAHIP_POSE_HYBRID_PARTS=3
  run heavy4_$rep BAHIP_HEAVY_SMALL_X2=4
  run heavy3_$rep BAHIP_HEAVY_SMALL_X2=3
  run heavy3p3_$rep BAHIP_HEAVY_SMALL_X2=3 BAHIP_POSE_HYBRID_PARTS=3
done
