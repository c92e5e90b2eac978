#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
{
for k in 1 2 3 4 6; do for pools in 2 3 4; do echo "== f32 eager_top $k pools $pools"; python3 -m bokego_amd.selfplay --games 512 --rollouts 400 --eager-top $k --pools $pools 2>&1 | grep -v amdgpu.ids | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['games_per_min']), d['seconds'], d['leaf_evals'], d['mean_batch'])"; done; done
for k in 8 16 24; do for pools in 3 4; do echo "== f16x2 eager_top $k pools $pools"; python3 -m bokego_amd.selfplay --games 512 --rollouts 400 --eager-top $k --pools $pools --precision f16x2 2>&1 | grep -v amdgpu.ids | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['games_per_min']), d['seconds'], d['leaf_evals'], d['mean_batch'])"; done; done
} > gpurun_out/r03_eager_top2.txt 2>&1
cat gpurun_out/r03_eager_top2.txt
