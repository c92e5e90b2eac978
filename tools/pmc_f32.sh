#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/pmc_f32; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd $ROOT
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30); rocprofv3 --pmc $C --output-format csv -d $OUT/$N -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-selfplay --precision f32 > $OUT/$N.log 2>&1
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob('$OUT/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'bk_leaf_eval_kernel' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items(): print(k, sum(v)/len(v), len(v))
PY
