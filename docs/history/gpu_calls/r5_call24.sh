#!/bin/bash
# round 5, call 24: full GPU suite + smoke on the round's last code state
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call24; mkdir -p $O
timeout -k 5 700 python -m pytest tests -q -m gpu 2>&1 | tail -6 > $O/gpu_tests.log
tail -3 $O/gpu_tests.log | cut -c1-300
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 5 300 python bench.py --no-cpu-baseline --no-extras | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['loop'], d['prepass']['iterations'])"
