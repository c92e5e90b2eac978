#!/bin/bash
# A/B of an environment switch on the 512-game generation, both precisions, alternating.  usage: tools/r03_probe15.sh VAR [reps]
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/probe15
VAR=${1:-BK_NO_LANES}
for i in $(seq 1 ${2:-3}); do
  for PREC in f16x2 f32; do
    for L in 0 1; do
      if [ $L = 1 ]; then export $VAR=1; else unset $VAR; fi
      timeout -k 10 120 python3 -m bokego_amd.selfplay --games 512 --precision $PREC > gpurun_out/probe15/o.out 2> gpurun_out/probe15/o.err || { echo FAILED; tail -5 gpurun_out/probe15/o.err; exit 1; }
      python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(f\"$PREC $VAR=$L: {d['seconds']:.3f} s  {d['games_per_min']:.0f} games/min  black {d['black_wins']:.0f}\")" gpurun_out/probe15/o.out
    done
  done
done
