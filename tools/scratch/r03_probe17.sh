#!/bin/bash
# repeatability of the 512-game generation in fresh processes (outliers = CPU-time throttling of the spinning worker team?)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/probe17
cat /sys/fs/cgroup/cpu.max 2>/dev/null; grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null
for PREC in f16x2 f32; do
  for T in ${1:-12}; do
    echo -n "$PREC threads $T:"
    for i in $(seq 1 ${2:-10}); do
      timeout -k 10 120 python3 -m bokego_amd.selfplay --games 512 --precision $PREC --threads $T > gpurun_out/probe17/o.out 2> gpurun_out/probe17/o.err || { echo FAILED; tail -5 gpurun_out/probe17/o.err; exit 1; }
      python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(f\" {d['seconds']:.3f}\", end='')" gpurun_out/probe17/o.out
    done
    echo
  done
done
grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null
