#!/bin/bash
# Runs on the GPU box: bench + rocprofv3 kernel trace + PMC passes (separate runs, as gpurun requires).
# usage: tools/profile_bench.sh <tag>
set -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
python3 bench.py --steps 50 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || { echo bench failed; tail -5 $OUT/bench.err; exit 1; }
cat $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-selfplay > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
for C in FETCH_SIZE WRITE_SIZE "SQ_BUSY_CYCLES SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-selfplay > $OUT/pmc_$N.log 2>&1 || { echo "pmc $C failed"; tail -3 $OUT/pmc_$N.log; }
done
find $OUT -name "*.csv" | head -40
