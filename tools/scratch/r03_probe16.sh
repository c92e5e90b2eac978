#!/bin/bash
# f16x2 self-play with the lanes: request-size limits (whole rounds of 3-board workgroups: 768 tasks each) and children per expansion
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/probe16
for E in 4 6; do
  for CAP in 0 764 1148 1532; do
    echo -n "eager_top $E task cap $CAP: "
    timeout -k 10 120 python3 -m bokego_amd.selfplay --games 512 --precision f16x2 --eager-top $E --task-cap $CAP > gpurun_out/probe16/o.out 2> gpurun_out/probe16/o.err || { echo FAILED; tail -5 gpurun_out/probe16/o.err; exit 1; }
    python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(f\"{d['seconds']:.3f} s  {d['games_per_min']:.0f} games/min  mean batch {d['mean_batch']:.0f}\")" gpurun_out/probe16/o.out
  done
done
