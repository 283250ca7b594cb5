#!/bin/bash
# round 5, call 28: what the intrinsics sweep waits for -- timing experiments (WRONG results on purpose: the first launch's duration counts):
# without the record stores, without the reservation atomics, without both
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp BADSLAM_RENDER_WORKERS=8
O=$GRAFT_REPO_ROOT/gpurun_out/r5_call28; mkdir -p $O
cd /tmp
for v in - nostore noreserve neither; do
  if [ "$v" = "-" ]; then unset BADSLAM_LIB_DIR; else export BADSLAM_LIB_DIR=$GRAFT_REPO_ROOT/badslam_amd/lib_variants/$v; fi
  timeout -k 5 120 rocprofv3 --kernel-trace --output-format csv -d $O/t_$v -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --intrinsics --steps 2 --warmup 0 > /dev/null 2> $O/$v.err
  f=$(find $O/t_$v -name '*kernel_trace.csv' | head -1)
  python - "$v" $f <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "intrinsics_accumulate_kernel" in r["Kernel_Name"]]
p = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "pose_accumulate_lds" in r["Kernel_Name"]]
print("%-10s intrinsics sweep launches (us): %s   | first pose sweep %s" % (sys.argv[1], [round(x) for x in d[:4]], [round(x) for x in p[:2]]))
PY
  rm -rf $O/t_$v
done
