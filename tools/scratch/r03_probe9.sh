#!/bin/bash
# one rank's share of configs[3] at 8 ranks (64 of the 512 games, 2 host threads as bench.py's plan gives a rank of 8):
# pools and children-per-expansion sweep, fp32 and f16x2.  usage: tools/r03_probe9.sh [precisions] [pools]
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/probe9
for PREC in ${1:-f32 f16x2}; do
  for P in ${2:-1 2 3 4}; do
    for E in 4 8; do
      echo -n "$PREC pools $P eager_top $E: "
      timeout -k 10 120 python3 -m bokego_amd.selfplay --games 512 --replay-shard 0/8 --pools $P --threads 2 --precision $PREC --eager-top $E > gpurun_out/probe9/${PREC}_${P}_$E.out 2> gpurun_out/probe9/${PREC}_${P}_$E.err
      rc=$?
      if [ $rc -ne 0 ]; then echo "FAILED rc=$rc"; tail -20 gpurun_out/probe9/${PREC}_${P}_$E.err; exit 1; fi
      python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(f\"{d['seconds']:.3f} s  {d['games_per_min']:.0f} games/min (x8 = {8*d['games_per_min']:.0f})  mean batch {d['mean_batch']:.0f}\")" gpurun_out/probe9/${PREC}_${P}_$E.out
    done
  done
done
