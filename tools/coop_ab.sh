# the last A/B run around the small-batch path on the GPU box (edit freely)
cd /root/repo
export PYTHONPATH=/root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_mcts.py -x -q -m gpu 2>&1 | tail -3
timeout -k 10 200 python tools/genmove_probe.py 40 f16x2 2>&1 | grep -v amdgpu | head -3
timeout -k 10 200 python tools/genmove_probe.py 40 2>&1 | grep -v amdgpu | head -3
