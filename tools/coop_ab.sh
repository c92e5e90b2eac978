# the last A/B run around the cooperative launch on the GPU box (edit freely)
set -e
cd /root/repo
export PYTHONPATH=/root/repo
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cooperative or overflow"
timeout -k 10 200 python tools/coop_probe.py
timeout -k 10 200 python tools/genmove_probe.py 40
