#!/bin/bash
# round 6, call 28: the hybrid geometry kernel at 4 wavefronts per SIMD (17 spilled registers) against 3 (no spills): emulated 8-rank share, alternating
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r6_call28; mkdir -p $O
for rep in 1 2 3; do
  for v in - hyb3; do
    if [ "$v" = "-" ]; then unset BADSLAM_LIB_DIR; else export BADSLAM_LIB_DIR=$PWD/badslam_amd/lib_variants/$v; fi
    timeout -k 5 300 python bench.py --emulate-world 8 --force-allreduce --no-cpu-baseline --no-extras > $O/emu8_${v}_$rep.json 2> $O/emu8_${v}_$rep.log
    python -c "import json; d=json.load(open('$O/emu8_${v}_$rep.json')); print('$v rep $rep:', round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['stage_ms_per_iteration'].items()})"
  done
done
