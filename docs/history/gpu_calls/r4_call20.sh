#!/bin/bash
# round 4, call 20: the round's artifacts -- full GPU suite, rocprofv3 passes of the default bench (kernel stats + PMC), the bench
# line with its extras and the CPU baseline, emulated shares, and the kernel stats + HBM counters of the intrinsics and PCG legs
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r4_call20; mkdir -p $O
timeout -k 5 400 python -m pytest tests -q -m gpu 2>&1 | tail -25 > $O/gpu_tests.log
tail -3 $O/gpu_tests.log
PASS_TIMEOUT=120 bash scripts/profile_round.sh r4_a > $O/profile_r4_a.log 2>&1
cp gpurun_out/prof_r4_a/pmc_per_kernel.json profiles/r4_a_pmc_per_kernel.json 2>/dev/null
unset BADSLAM_RENDER_WORKERS
timeout -k 5 400 python bench.py > $O/bench_final.json 2> $O/bench_final.err
for w in 8 4 2; do
  timeout -k 5 100 python bench.py --emulate-world $w --force-allreduce --no-cpu-baseline --no-extras > $O/bench_emu$w.json 2> $O/bench_emu$w.err
done
export BADSLAM_RENDER_WORKERS=32
CAL_FROM=$GRAFT_REPO_ROOT/gpurun_out/prof_r4_a PASSES=basic PASS_TIMEOUT=120 bash scripts/profile_round.sh r4_intr --intrinsics > $O/profile_r4_intr.log 2>&1
CAL_FROM=$GRAFT_REPO_ROOT/gpurun_out/prof_r4_a PASSES=basic PASS_TIMEOUT=120 bash scripts/profile_round.sh r4_pcg --pcg > $O/profile_r4_pcg.log 2>&1
python - <<'PY'
import json
for f in ["bench_final","bench_emu8","bench_emu4","bench_emu2"]:
    try:
        d=json.load(open(f"gpurun_out/r4_call20/{f}.json"))
        print(f, round(d["value"],1), round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["stage_ms_per_iteration"].items()})
    except Exception as e: print(f, "failed", e)
PY
