#!/bin/bash
# round 4, call 2: bisect of the round-3 LDS anomaly on the commit that had it (2a0bfbd: v0 as it was, v1 + pin, v2 + batch word
# moved behind the table, v3 both); the full GPU suite with the implicit Morton reorder; the default bench line with the new extras
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_call2; mkdir -p $O
for v in v0 v1 v2 v3; do
  echo "== bisect $v"
  (cd _bisect/$v && timeout 600 python -m pytest tests/test_gpu_kernels_vs_oracle.py tests/test_gpu_scale_parity.py -q -m gpu -k "lds" 2>&1 | tail -12) | tee $O/bisect_$v.log
done
echo "== full gpu suite"; timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -15 | tee $O/gpu_tests.log
echo "== bench default"; timeout 900 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err; python - <<'PY'
import json,os
d=json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/r4_call2/bench_default.json")))
print("value", d["value"], "ms", d["ms_per_step"], d["stage_ms_per_iteration"])
for k in ("unsorted_ba_iterations_per_s","drop_in","cold_start","intrinsics","pcg"):
    print(k, d.get(k))
print("roofline", {k: d["roofline"][k] for k in ("achieved","frac","avg_launch_ms","launches","keyframes_per_launch")})
PY
