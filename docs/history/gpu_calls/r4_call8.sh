#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_call8; mkdir -p $O
(cd _bisect/v13 && timeout 300 python diag_hb.py 2>&1 | grep -v "^item" | head -40) | tee $O/diag_v13.log
scripts/experiments/bin/smulk_check | tee $O/smulk.log
echo "== device loop with fused iteration begin"; timeout 600 python -m pytest tests/test_gpu_device_loop.py tests/test_gpu_directba_vs_oracle.py -q -m gpu 2>&1 | tail -3
for dl in 1 0 1 0; do
  BAHIP_DEVICE_LOOP=$dl timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms_per_iteration']
print('device_loop=$dl: %.1f it/s  %.3f ms/iter | ' % (d['value'], d['ms_per_step']) + '  '.join('%s %.3f' % (k, v) for k, v in s.items()))" | tee -a $O/ab.log
done
timeout 300 python bench.py --no-cpu-baseline --no-extras --emulate-world 8 --force-allreduce 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms_per_iteration']
print('world 8: %.3f ms/iter | ' % d['ms_per_step'] + '  '.join('%s %.3f' % (k, v) for k, v in s.items()))" | tee -a $O/ab.log
