#!/bin/bash
# scripts/pmc_pass.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...] -- PMC-only passes over bench.py (5 steps)
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
i=0
for ctrs in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$OUT/p$i" -- python $REPO/bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2> "$OUT/p$i.log"
done
cd $REPO
python - "$OUT" <<'PY'
import sys,glob,csv,collections,json
out=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(lambda:[0.0,0]))
for f in glob.glob(out+'/p*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name'].split('(')[0].replace('void ','').replace('bahip::','')
        a=acc[n][r['Counter_Name']]; a[0]+=float(r['Counter_Value']); a[1]+=1
res={k:{c:v[0]/v[1] for c,v in d.items()} for k,d in acc.items()}
json.dump(res,open(out+'/pmc.json','w'),indent=1,sort_keys=True)
for k in ('pose_accumulate_kernel<true, true>','geometry_kernel<true, true>','activation_kernel'):
    if k in res: print(k, json.dumps(res[k],sort_keys=True))
PY
find "$OUT" -name '*.csv' -size +1M -delete
