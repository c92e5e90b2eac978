#!/bin/bash
# process-level stress of the self-play CLI (a run of tools/r03_probe9.sh once ended with empty output and its stderr
# thrown away): up to N fresh processes, stop at the first that fails or prints nothing, keep its stderr
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/probe10
N=${1:-24}
for i in $(seq 1 $N); do
  P=$((1 + i % 4)); PREC=f16x2; [ $((i % 3)) -eq 0 ] && PREC=f32
  timeout -k 10 120 python3 -X faulthandler -m bokego_amd.selfplay --games 512 --replay-shard 0/8 --pools $P --threads 2 --precision $PREC --eager-top 4 > gpurun_out/probe10/run.out 2> gpurun_out/probe10/run.err
  rc=$?
  if [ $rc -ne 0 ] || ! grep -q games_per_min gpurun_out/probe10/run.out; then
    echo "run $i ($PREC pools $P): rc=$rc"; cp gpurun_out/probe10/run.err gpurun_out/probe10/failed_$i.err; tail -40 gpurun_out/probe10/run.err; exit 1
  fi
  echo "run $i ($PREC pools $P) ok"
done
