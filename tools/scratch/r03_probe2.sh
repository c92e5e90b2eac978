#!/bin/bash
# round-3 GPU probe 2: full suite on the final build, default bench, stamps of both fp32 forms, launcher rehearsal, match figures
set -o pipefail
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/r03_t4.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03_t4.log; tail -3 gpurun_out/r03_t4.log
python3 bench.py > gpurun_out/r03_bench_c.json 2> gpurun_out/r03_bench_c.err; echo "bench rc=$?"
BK_LIB_PATH=bokego_amd/libbokego_amd_diag.so BK_LIB_ANY_ABI=1 python3 tools/stamp_coop.py 62 > gpurun_out/r03_stamps_coop.txt 2>&1; echo "stamps coop rc=$?"
BK_LIB_PATH=bokego_amd/libbokego_amd_diag.so BK_LIB_ANY_ABI=1 python3 tools/stamp_profile.py > gpurun_out/r03_stamps_f32.txt 2>&1; echo "stamps f32 rc=$?"
BK_BENCH_BACKEND=gloo BK_BENCH_DEVICE=0 python3 bench.py --gpus 2 --steps 10 --warmup 3 > gpurun_out/r03_rehearsal2.json 2> gpurun_out/r03_rehearsal2.err; echo "rehearsal rc=$?"
python3 tools/coop_probe.py > gpurun_out/r03_coop_default.txt 2>&1
python3 tools/genmove_probe.py 80 > gpurun_out/r03_genmove80.txt 2>&1; tail -3 gpurun_out/r03_genmove80.txt
