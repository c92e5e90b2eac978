#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
{
for k in 0 8 12 16 4; do echo "== genmove eager_top $k"; BK_EAGER_TOP=$k python3 tools/genmove_probe.py 80 2>&1 | grep -E "ms/move|evaluator calls|value evals"; done
for k in 0 8 12 16 4; do for p in f32 f16x2; do echo "== selfplay 512x400 eager_top $k $p"; python3 -m bokego_amd.selfplay --games 512 --rollouts 400 --eager-top $k --precision $p 2>&1 | grep -v amdgpu.ids; done; done
echo "== selfplay 64 games (a rank's share at 8 GPUs)"; for k in 0 8; do python3 -m bokego_amd.selfplay --games 64 --rollouts 400 --eager-top $k 2>&1 | grep -v amdgpu.ids; done
} > gpurun_out/r03_eager_top.txt 2>&1
cat gpurun_out/r03_eager_top.txt | cut -c1-330
