#!/bin/bash
# per-dispatch FETCH_SIZE / WRITE_SIZE of the B = 4,096 fp32 launch (cold first launch against steady state)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/probe8
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 bench.py --pmc-child --batch 4096 --precision f32 > $OUT/$C.log 2>&1 || { echo "$C failed"; tail -3 $OUT/$C.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "probe8")
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(out, c, "**", "*_counter_collection.csv"), recursive=True):
        rows = [x for x in csv.DictReader(open(f)) if "bk_leaf_eval_kernel" in x["Kernel_Name"]]
        rows.sort(key=lambda x: int(x["Start_Timestamp"]))
        print(c, [(x["Grid_Size"], round(float(x["Counter_Value"]) / 1024, 1)) for x in rows])
PY
