#!/bin/bash
# round 4, call 3: bisect v4 (the failing commit + ONLY the 2^40 -> 2^52 range change); device-driven loop; e2e VGA parity; full suite;
# A/B of the device-driven loop against the host-driven one; emulated shares
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_call3; mkdir -p $O
echo "== bisect v4"; (cd _bisect/v4 && timeout 600 python -m pytest tests/test_gpu_kernels_vs_oracle.py tests/test_gpu_scale_parity.py -q -m gpu -k "lds" 2>&1 | tail -5) | tee $O/bisect_v4.log
echo "== device loop + e2e"; timeout 900 python -m pytest tests/test_gpu_device_loop.py tests/test_gpu_e2e_vga.py -q -m gpu -s 2>&1 | tail -25 | tee $O/device_loop_e2e.log
echo "== full gpu suite"; timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -15 | tee $O/gpu_tests.log
echo "== A/B device loop"
for r in 1 2; do for dl in 1 0; do
  BAHIP_DEVICE_LOOP=$dl timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms_per_iteration']
print('device_loop=$dl: %.1f it/s  %.3f ms/iter | ' % (d['value'], d['ms_per_step']) + '  '.join('%s %.3f' % (k, v) for k, v in s.items()), 'launch', round(d['roofline']['avg_launch_ms'],4), d['roofline']['launches'], 'R', d['config']['pose_gn_rounds_per_iteration'])" | tee -a $O/ab_device_loop.log
done; done
echo "== emulate"
for w in 8 4 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-extras --emulate-world $w --force-allreduce 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms_per_iteration']
print('world $w: %.3f ms/iter | ' % d['ms_per_step'] + '  '.join('%s %.3f' % (k, v) for k, v in s.items()), d['roofline']['launches_by_form'])" | tee -a $O/emulate.log
done
