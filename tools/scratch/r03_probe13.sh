#!/bin/bash
# one rank's share of configs[3] at W ranks: host threads sweep, fp32, default pools.  usage: tools/r03_probe13.sh "W:threads..." ...
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/probe13
for SPEC in "$@"; do
  W=${SPEC%%:*}
  for T in ${SPEC#*:}; do
    echo -n "world $W threads $T: "
    timeout -k 10 120 python3 -m bokego_amd.selfplay --games 512 --replay-shard 0/$W --threads $T > gpurun_out/probe13/o.out 2> gpurun_out/probe13/o.err || { echo FAILED; tail -5 gpurun_out/probe13/o.err; exit 1; }
    python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(f\"{d['seconds']:.3f} s  x$W = {$W*d['games_per_min']:.0f} games/min  mean batch {d['mean_batch']:.0f}\")" gpurun_out/probe13/o.out
  done
done
