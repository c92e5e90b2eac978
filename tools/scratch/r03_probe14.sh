#!/bin/bash
# 64-game share (one rank of eight), fp32, 4 threads: request size limits around the cooperative kernel's steps (63 / 85 / 127 tasks)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/probe14
for P in 2 3; do
  for CAP in 0 63 85 127; do
    echo -n "pools $P task cap $CAP: "
    timeout -k 10 120 python3 -m bokego_amd.selfplay --games 512 --replay-shard 0/8 --pools $P --threads 4 --task-cap $CAP > gpurun_out/probe14/o.out 2> gpurun_out/probe14/o.err || { echo FAILED; tail -5 gpurun_out/probe14/o.err; exit 1; }
    python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(f\"{d['seconds']:.3f} s  x8 = {8*d['games_per_min']:.0f} games/min  mean batch {d['mean_batch']:.0f}\")" gpurun_out/probe14/o.out
  done
done
