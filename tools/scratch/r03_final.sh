#!/bin/bash
# round-3 final artefacts: profile_bench (bench + kernel trace + PMC passes), self-play kernel stats, stamps are from probe 2
set -o pipefail
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/profile_bench.sh r03 "f32 f16x2" > gpurun_out/r03_profile.log 2>&1; tail -3 gpurun_out/r03_profile.log
for P in f32 f16x2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_selfplay_$P -- python3 -m bokego_amd.selfplay --games 512 --rollouts 400 --precision $P > gpurun_out/r03_selfplay_$P.log 2>&1; echo "selfplay trace $P rc=$?"
done
python3 tools/host_tree_bench.py 256 4 12 > gpurun_out/r03_host_tree3.txt 2>&1; for t in 1 4 8 12 16; do python3 tools/host_tree_bench.py 256 4 $t; done >> gpurun_out/r03_host_tree3.txt 2>&1
