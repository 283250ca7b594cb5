#!/bin/bash
# round 4, call 16: rounds-per-phase policy of the device loop, register-resident LDLT in the solve kernel, wave-parallel creation
# filter, tile-culled lifecycle sweeps: full GPU suite, default bench, emulated 8-rank share, kernel stats
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_call16; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $O/gpu_tests.log
tail -3 $O/gpu_tests.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
for w in 8 4 2; do
  timeout 200 python bench.py --emulate-world $w --force-allreduce --no-cpu-baseline --no-extras > $O/bench_emu$w.json 2> $O/bench_emu$w.err
done
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_emu8 -- python $GRAFT_REPO_ROOT/bench.py --emulate-world 8 --force-allreduce --no-cpu-baseline --no-extras > $O/emu8_prof.json 2> $O/emu8_prof.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json,glob
for f in ["bench_default","bench_emu8","bench_emu4","bench_emu2"]:
    try:
        d=json.load(open(f"gpurun_out/r4_call16/{f}.json"))
        print(f, d["value"], d["ms_per_step"], d["stage_ms_per_iteration"], d.get("drop_in",{}).get("ms_per_call"), d.get("drop_in",{}).get("lifecycle_and_end_tasks_ms_per_call"), d.get("cold_start",{}).get("ba_iterations_per_s"))
    except Exception as e: print(f, "failed", e)
PY
