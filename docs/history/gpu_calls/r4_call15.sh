#!/bin/bash
# round 4, call 15: kernel traces -- the bench with its extras (where does the lifecycle of the drop_in leg go?), the emulated 8-rank
# share (what fills the 0.1 ms between the kernels?), and that share with the host-driven loop / one round queued ahead
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_call15; mkdir -p $O
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_emu8 -- python $GRAFT_REPO_ROOT/bench.py --emulate-world 8 --force-allreduce --no-cpu-baseline --no-extras > $O/emu8_prof.json 2> $O/emu8_prof.err
cd $GRAFT_REPO_ROOT
for v in "BAHIP_DEVICE_LOOP=0" "BAHIP_POSE_ROUNDS_AHEAD=1" "BAHIP_POSE_ROUNDS_AHEAD=3" "X=1"; do
  echo "== $v" >> $O/emu8_variants.log
  env $v timeout 200 python bench.py --emulate-world 8 --force-allreduce --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stage_ms_per_iteration'], d['config'].get('pose_gn_rounds_per_iteration'))" >> $O/emu8_variants.log
done
cat $O/emu8_variants.log
find $O -name "*.csv" | xargs ls -la
