#!/bin/bash
# round 4, call 17: phase-ending solve opens the next iteration; sorted record reduction of the intrinsics step (both forms);
# full GPU suite, default bench, emulated shares, intrinsics leg with either reduction
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_call17; mkdir -p $O
timeout -k 5 600 python -m pytest tests -q -m gpu -x 2>&1 | tail -25 > $O/gpu_tests.log
tail -4 $O/gpu_tests.log
timeout -k 5 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
for w in 8 4 2; do
  timeout -k 5 120 python bench.py --emulate-world $w --force-allreduce --no-cpu-baseline --no-extras > $O/bench_emu$w.json 2> $O/bench_emu$w.err
done
for f in 0 1; do
  BAHIP_INTR_REDUCE_FORM=$f timeout -k 5 120 python bench.py --intrinsics --steps 5 --no-cpu-baseline --no-extras > $O/bench_intr_form$f.json 2> $O/bench_intr_form$f.err
done
python - <<'PY'
import json
for f in ["bench_default","bench_emu8","bench_emu4","bench_emu2","bench_intr_form0","bench_intr_form1"]:
    try:
        d=json.load(open(f"gpurun_out/r4_call17/{f}.json"))
        print(f, round(d["value"],1), round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["stage_ms_per_iteration"].items()}, (d.get("drop_in") or {}).get("ms_per_call"), (d.get("intrinsics") or {}))
    except Exception as e: print(f, "failed", e)
PY
