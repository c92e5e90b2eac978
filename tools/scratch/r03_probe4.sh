#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
run() { echo "== speculate $1 rows $2 tasks $3 steps $4"; BK_SPECULATE=$1 BK_SPECULATE_ROWS=$2 BK_REQUEST_TASKS=$3 BK_REQUEST_STEPS=$4 python3 tools/genmove_probe.py 80 2>&1 | grep -E "ms/move|evaluator calls|value evals"; }
{ run 50 80 64 64,80,128; run 50 128 128 128,256,512; run 30 128 128 128,256,512; run 20 128 128 128,256,512; run 50 128 64 64,128,256; run 30 128 64 64,128,256; run 40 80 64 64,80,128; run 50 80 64 64,80,128; } > gpurun_out/r03_request_steps_ab2.txt 2>&1
cat gpurun_out/r03_request_steps_ab2.txt
