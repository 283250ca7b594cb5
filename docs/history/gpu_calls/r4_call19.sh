#!/bin/bash
# round 4, call 19: static deal of the tiles for short later-round lists: pose parity tests, timing at full size and on shares
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_call19; mkdir -p $O
timeout -k 5 300 python -m pytest tests/test_gpu_scale_parity.py tests/test_gpu_device_loop.py tests/test_gpu_sharded_loopback.py tests/test_gpu_pose_vs_oracle.py -q -m gpu -x -k "pose or loop or shard" 2>&1 | tail -6 > $O/gpu_tests.log
tail -3 $O/gpu_tests.log
for run in 1 2; do
timeout -k 5 120 python bench.py --no-cpu-baseline --no-extras > $O/bench_default_$run.json 2> $O/bench_default.err
done
for w in 8 4 2; do
  timeout -k 5 120 python bench.py --emulate-world $w --force-allreduce --no-cpu-baseline --no-extras > $O/bench_emu$w.json 2> $O/bench_emu$w.err
done
python - <<'PY'
import json
for f in ["bench_default_1","bench_default_2","bench_emu8","bench_emu4","bench_emu2"]:
    try:
        d=json.load(open(f"gpurun_out/r4_call19/{f}.json"))
        print(f, round(d["value"],1), round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["stage_ms_per_iteration"].items()})
    except Exception as e: print(f, "failed", e)
PY
