# the last A/B run around the small-batch path on the GPU box (edit freely)
cd /root/repo
export PYTHONPATH=/root/repo
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mcts.py -x -q -m gpu -k "cooperative or config3" 2>&1 | tail -3
timeout -k 10 200 python tools/coop_probe.py 2>&1 | grep -v amdgpu
for p in f32 f16x2; do timeout -k 10 200 python tools/genmove_probe.py 40 $p 2>&1 | grep -v amdgpu | head -3; done
for s in 0 50 70 85; do BK_SPECULATE=$s BK_SPECULATE_ROWS=128 timeout -k 10 200 python tools/genmove_probe.py 40 2>&1 | grep -v amdgpu | head -3; done
