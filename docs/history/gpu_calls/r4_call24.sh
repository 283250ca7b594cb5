#!/bin/bash
# round 4, call 24: the final state's artifacts (tag r4_b): full GPU suite, smoke, rocprofv3 passes of the default bench, the bench
# line with extras and CPU baseline, emulated shares
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_call24; mkdir -p $O
timeout -k 5 400 python -m pytest tests -q -m gpu 2>&1 | tail -25 > $O/gpu_tests.log
tail -3 $O/gpu_tests.log
timeout -k 5 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
BADSLAM_RENDER_WORKERS=32 PASS_TIMEOUT=120 bash scripts/profile_round.sh r4_b > $O/profile_r4_b.log 2>&1
cp gpurun_out/prof_r4_b/pmc_per_kernel.json profiles/r4_b_pmc_per_kernel.json 2>/dev/null
timeout -k 5 400 python bench.py > $O/bench_final.json 2> $O/bench_final.err
for w in 8 4 2; do
  timeout -k 5 100 python bench.py --emulate-world $w --force-allreduce --no-cpu-baseline --no-extras > $O/bench_emu$w.json 2> $O/bench_emu$w.err
done
python - <<'PY'
import json
for f in ["bench_final","bench_emu8","bench_emu4","bench_emu2"]:
    try:
        d=json.load(open(f"gpurun_out/r4_call24/{f}.json"))
        print(f, round(d["value"],1), round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["stage_ms_per_iteration"].items()}, (d.get("drop_in") or {}).get("ms_per_call"))
    except Exception as e: print(f, "failed", e)
PY
