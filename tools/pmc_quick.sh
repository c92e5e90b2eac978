#!/bin/bash
# quick L2 / fabric counters of the default bench kernel (GPU box): tools/pmc_quick.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/pmcq_${1:-x}; mkdir -p $OUT; export TMPDIR=/tmp; cd $ROOT
for C in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_'); rocprofv3 --pmc $C --output-format csv -d $OUT/$N -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/$N.log 2>&1
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob('$OUT/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'bk_leaf' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items(): print(k, sum(v)/len(v))
PY
