# the last A/B run around the cooperative launch on the GPU box (edit freely)
set -e
cd /root/repo
export PYTHONPATH=/root/repo
timeout -k 10 200 python tools/genmove_probe.py 40
timeout -k 10 600 python -m pytest tests/test_gpu_mcts.py tests/test_gpu_selfplay.py -x -q -m gpu
