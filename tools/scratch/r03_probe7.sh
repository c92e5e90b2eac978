#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
{
for th in 4 6 8 12 16; do for cfg in "f32 4 2" "f16x2 4 3" "f16x2 8 3" "f16x2 4 2"; do set -- $cfg; echo "== threads $th $1 eager_top $2 pools $3"; python3 -m bokego_amd.selfplay --games 512 --rollouts 400 --eager-top $2 --precision $1 --pools $3 --threads $th 2>&1 | grep -v amdgpu.ids | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['games_per_min']), round(d['seconds'],3))"; done; done
echo "== 64 games"; for th in 1 2 4 8; do python3 -m bokego_amd.selfplay --games 64 --rollouts 400 --threads $th 2>&1 | grep -v amdgpu.ids | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print($th, round(d['games_per_min']), round(d['seconds'],3))"; done
} > gpurun_out/r03_threads.txt 2>&1
cat gpurun_out/r03_threads.txt
