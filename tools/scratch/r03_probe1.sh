#!/bin/bash
# round-3 GPU probe 1: tests, bench, cooperative-form variants, LDS-read experiment build, roctx marker trace
set -o pipefail
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/r03_t3.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03_t3.log; tail -3 gpurun_out/r03_t3.log
python3 bench.py > gpurun_out/r03_bench_a.json 2> gpurun_out/r03_bench_a.err; echo "bench rc=$?"
python3 tools/coop_probe.py > gpurun_out/r03_coop_default.txt 2>&1
BK_COOP=8 python3 tools/coop_probe.py > gpurun_out/r03_coop8.txt 2>&1
BK_COOP=6 python3 tools/coop_probe.py > gpurun_out/r03_coop6.txt 2>&1
for i in 1 2; do python3 tools/exp_rate.py; BK_LIB_PATH=bokego_amd/libbokego_amd_exp1.so BK_LIB_ANY_ABI=1 python3 tools/exp_rate.py; done > gpurun_out/r03_exp1.txt 2>&1
BK_ROCTX=1 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d gpurun_out/r03_roctx -- python3 tools/genmove_probe.py > gpurun_out/r03_roctx.log 2>&1; echo "roctx rc=$?"
find gpurun_out/r03_roctx -name "*.csv" | head; 
