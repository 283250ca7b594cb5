#!/bin/bash
# round 6, call 44: the frusta of a later round's short list in LDS: pose parity tests (exact and fast), the device loop, one GPU and the emulated share, the later rounds' launch durations
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r6_call44; mkdir -p $O
timeout -k 5 1200 python -m pytest tests/test_gpu_kernels_vs_oracle.py tests/test_gpu_device_loop.py tests/test_gpu_scale_parity.py tests/test_gpu_sharded_loopback.py tests/test_gpu_directba_vs_oracle.py tests/test_gpu_fast_flavour.py -q -m gpu -x 2>&1 | tail -3 | cut -c1-300
for rep in 1 2 3; do
  timeout -k 5 300 python bench.py --no-cpu-baseline --no-extras > $O/w1_$rep.json 2> /dev/null
  timeout -k 5 300 python bench.py --emulate-world 8 --force-allreduce --no-cpu-baseline --no-extras > $O/w8_$rep.json 2> /dev/null
  python -c "import json; a=json.load(open('$O/w1_$rep.json')); b=json.load(open('$O/w8_$rep.json')); print('rep $rep: one GPU', round(a['value'],1), 'it/s', 'pose stage', round(a['stage_ms_per_iteration']['pose_accumulate'],4), '; world 8 share', round(b['ms_per_step'],4), 'ms, pose stage', round(b['stage_ms_per_iteration']['pose_accumulate'],4))"
done
