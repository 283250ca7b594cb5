#!/bin/bash
# round 6, call 24: artifact set r6_b -- full GPU suite, the default bench line, counter evidence of the three legs, emulated shares
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
TAG=${TAG:-r6_c}
O=$GRAFT_REPO_ROOT/gpurun_out/${CALL:-r6_call24}; mkdir -p $O
timeout -k 5 1500 python -m pytest tests -q -m gpu --durations=5 2>&1 | tail -40 > $O/gpu_tests.log
tail -3 $O/gpu_tests.log | cut -c1-300
timeout -k 5 600 python bench.py > $O/bench.json 2> $O/bench.log
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print("value", round(d["value"],1), "exact", round(d["exact_arithmetic"]["ba_iterations_per_s"],1), "cold", round(d["cold_start"]["ba_iterations_per_s"],1), "unsorted", round(d["unsorted_ba_iterations_per_s"],1))
print("drop_in", d["drop_in"]["ms_per_call"], d["drop_in"]["ms_per_call_iterations_only"], "pcg", d["pcg"]["outer_iterations_per_s"], d["pcg"]["inner_steps_per_outer_iteration"], "intr", d["intrinsics"]["BA_intrinsics_optimization_ms_per_iteration"])
PY
BADSLAM_RENDER_WORKERS=8 timeout -k 5 1500 bash scripts/profile_all.sh $TAG > $O/profile_all.log 2>&1
grep -h '^cp ' $O/profile_all.log | sed "s#$GRAFT_REPO_ROOT/##g" > $O/cp_lines.sh
for W in 8 4 2; do
  timeout -k 5 300 python bench.py --emulate-world $W --force-allreduce --no-cpu-baseline --no-extras > $O/bench_emulated_world$W.json 2> $O/emu$W.log
  python -c "import json; d=json.load(open('$O/bench_emulated_world$W.json')); print('world $W:', round(d['ms_per_step'],4), 'ms per iteration', d.get('exchange',{}).get('link_budget'))"
done
