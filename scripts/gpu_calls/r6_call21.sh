#!/bin/bash
# round 6, call 21: the hybrid shape of the geometry step: parity of the shapes, then the emulated 8-rank share with each shape (alternating runs)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r6_call21; mkdir -p $O
timeout -k 5 900 python -m pytest tests/test_gpu_device_loop.py tests/test_gpu_kernels_vs_oracle.py tests/test_gpu_sharded_loopback.py -q -m gpu -x 2>&1 | tail -5 | cut -c1-300
for rep in 1 2; do
  for shape in 4 5 1; do
    BAHIP_TILE_WAVES=$shape timeout -k 5 300 python bench.py --emulate-world 8 --force-allreduce --no-cpu-baseline --no-extras > $O/emu8_shape${shape}_$rep.json 2> $O/emu8_shape${shape}_$rep.log
    python - <<PY
import json
d=json.load(open("$O/emu8_shape${shape}_$rep.json"))
st=d.get("stage_ms_per_iteration",{})
print("world 8 shape $shape rep $rep:", round(d["ms_per_step"],4), "ms per iteration;", {k: round(v,4) for k,v in st.items() if isinstance(v,(int,float))})
PY
  done
done
for W in 4 2; do
  for shape in 0 5; do
    BAHIP_TILE_WAVES=$shape timeout -k 5 300 python bench.py --emulate-world $W --force-allreduce --no-cpu-baseline --no-extras > $O/emu${W}_shape$shape.json 2> $O/emu${W}_shape$shape.log
    python -c "import json; d=json.load(open('$O/emu${W}_shape$shape.json')); print('world $W shape $shape:', round(d['ms_per_step'],4))"
  done
done
