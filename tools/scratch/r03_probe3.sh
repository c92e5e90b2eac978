#!/bin/bash
# A/B of the request-size steps of the native tree (bk_search_params.request_tasks / request_steps) on 80-move games, alternating
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
for i in 1 2 3; do
  for v in 0 64; do echo "== BK_REQUEST_TASKS=$v run $i"; BK_REQUEST_TASKS=$v python3 tools/genmove_probe.py 80 2>&1 | grep -v amdgpu.ids; done
done > gpurun_out/r03_request_steps_ab.txt 2>&1
for v in "50 80 64" "30 80 64" "50 64 64" "70 80 64"; do set -- $v; echo "== speculate $1 rows $2 tasks $3"; BK_SPECULATE=$1 BK_SPECULATE_ROWS=$2 BK_REQUEST_TASKS=$3 python3 tools/genmove_probe.py 80 2>&1 | grep -v amdgpu.ids; done >> gpurun_out/r03_request_steps_ab.txt 2>&1
grep -E "==|ms/move|evaluator calls" gpurun_out/r03_request_steps_ab.txt
