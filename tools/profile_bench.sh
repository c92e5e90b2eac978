#!/bin/bash
# Runs on the GPU box: bench + rocprofv3 kernel trace + PMC passes (separate runs, as gpurun requires).
# usage: tools/profile_bench.sh <tag> [precisions="f32 f16x2"]
# The kernel trace covers the default bench (fp32 headline + nested f16x2 block: both kernels in one csv); the PMC
# passes run one precision each (--no-f16x2), so every counter row belongs to one kernel.
set -o pipefail
TAG=${1:-r02}
PRECS=${2:-"f32 f16x2"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { echo bench failed; tail -5 $OUT/bench.err; exit 1; }
cat $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 3 --sustain 0.5 --no-cpu-baseline --no-selfplay --no-live-pmc > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
for P in $PRECS; do
  for C in FETCH_SIZE WRITE_SIZE "SQ_BUSY_CYCLES SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${P}_$N -- python3 bench.py --precision $P --no-f16x2 --steps 5 --warmup 2 --sustain 0 --no-cpu-baseline --no-selfplay --no-live-pmc > $OUT/pmc_${P}_$N.log 2>&1 || { echo "pmc $P $C failed"; tail -3 $OUT/pmc_${P}_$N.log; }
    echo "pmc $P $N done"
  done
done
find $OUT -name "*.csv" | wc -l
