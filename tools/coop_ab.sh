set -e
cd /root/repo
export PYTHONPATH=/root/repo
BK_COOP=0 timeout -k 10 200 python tools/ab_bits.py dump /tmp/nocoop.npz
timeout -k 10 200 python tools/ab_bits.py dump /tmp/coop.npz
python tools/ab_bits.py cmp /tmp/nocoop.npz /tmp/coop.npz
timeout -k 10 200 python tools/coop_probe.py
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cooperative"
