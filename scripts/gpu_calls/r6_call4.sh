#!/bin/bash
# round 6, call 4: tile timelines of the emulated 8-rank share (geometry and pose sweeps), exact flavour; kernel trace of the share;
# new tests (CUDABuffer async, clamped fused append)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=32
O=$GRAFT_REPO_ROOT/gpurun_out/r6_call4; mkdir -p $O/tl8 $O/tl1
timeout -k 5 300 python -m pytest tests/test_gpu_directba_cpp.py tests/test_gpu_lifecycle_stages.py tests/test_gpu_intrinsics_pcg_vs_oracle.py -q -m gpu 2>&1 | tail -5
BADSLAM_LIB_DIR=$PWD/badslam_amd/lib_variants/timeline BAHIP_TIMELINE_DIR=$O/tl8 timeout -k 5 200 python bench.py --emulate-world 8 --force-allreduce --no-extras --no-cpu-baseline > $O/bench_tl8.json 2>/dev/null
python scripts/tile_timeline.py $O/tl8 | tee $O/timeline_world8.txt
for w in 8 4 2; do
  timeout -k 5 200 python bench.py --emulate-world $w --force-allreduce --no-extras --no-cpu-baseline 2>/dev/null > $O/emu$w.json
  python -c "
import json
d=json.load(open('$O/emu$w.json')); print('world $w exact', round(d['ms_per_step'],4), d['stage_ms_per_iteration'])"
  BENCH_ARITHMETIC=fast timeout -k 5 200 python bench.py --emulate-world $w --force-allreduce --no-extras --no-cpu-baseline 2>/dev/null > $O/emu${w}_fast.json
  python -c "
import json
d=json.load(open('$O/emu${w}_fast.json')); print('world $w fast ', round(d['ms_per_step'],4), d['stage_ms_per_iteration'])"
done
cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $GRAFT_REPO_ROOT/bench.py --emulate-world 8 --force-allreduce --no-extras --no-cpu-baseline > $O/trace.log 2>&1
python - <<'PY'
import csv, glob, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r6_call4"
f=glob.glob(O+"/trace/**/*kernel_trace.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
tail=rows[-40:]
prev=None
with open(O+"/emu8_kernel_timeline.txt","w") as out:
    for r in tail:
        s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
        gap=(s-prev)/1e3 if prev else 0
        line=f"{gap:8.1f} us gap | {(e-s)/1e3:8.1f} us | {r['Kernel_Name'][:60]}"
        print(line); out.write(line+"\n")
        prev=e
PY
find $O/trace -name '*.csv' -size +1M -delete
