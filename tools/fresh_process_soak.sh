#!/bin/bash
# N fresh self-play processes, one after the other (64-game shards, 1..4 pools, varying thread counts): the round-3 worker-team bug showed as 2 of 37 such
# processes dying in the first native call after a parallel region.  usage: tools/fresh_process_soak.sh [n=40]
N=${1:-40}; ok=0; bad=0
for i in $(seq 1 $N); do
  T=$(( (i % 4) * 3 + 2 )); P=$(( i % 4 + 1 ))
  if timeout -k 5 120 python -m bokego_amd.selfplay --games 64 --rollouts 200 --threads $T --pools $P > /tmp/fps_$$.log 2>&1; then ok=$((ok+1)); else bad=$((bad+1)); echo "process $i (threads $T pools $P) failed: $(tail -2 /tmp/fps_$$.log)"; fi
done
echo "fresh self-play processes: $ok ok, $bad failed"
