#!/bin/bash
# round 6, call 40: only the awaited solve launch of the device-driven loop publishes to host memory: tests, then one GPU and the emulated 8-rank share
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r6_call40; mkdir -p $O
timeout -k 5 900 python -m pytest tests/test_gpu_device_loop.py tests/test_gpu_directba_vs_oracle.py tests/test_gpu_sharded_loopback.py tests/test_gpu_scale_parity.py -q -m gpu -x 2>&1 | tail -4 | cut -c1-300
for rep in 1 2 3; do
  timeout -k 5 300 python bench.py --no-cpu-baseline --no-extras > $O/w1_$rep.json 2> $O/w1_$rep.log
  timeout -k 5 300 python bench.py --emulate-world 8 --force-allreduce --no-cpu-baseline --no-extras > $O/w8_$rep.json 2> $O/w8_$rep.log
  python -c "import json; a=json.load(open('$O/w1_$rep.json')); b=json.load(open('$O/w8_$rep.json')); print('rep $rep: one GPU', round(a['value'],1), 'it/s', round(a['ms_per_step'],4), 'ms; world 8 share', round(b['ms_per_step'],4), 'ms; solve stage', round(a['stage_ms_per_iteration']['pose_solve'],4), round(b['stage_ms_per_iteration']['pose_solve'],4))"
done
