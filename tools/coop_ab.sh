set -e
cd /root/repo
export PYTHONPATH=/root/repo
BK_COOP=0 timeout -k 10 200 python tools/ab_bits.py dump /tmp/nocoop.npz
timeout -k 10 200 python tools/ab_bits.py dump /tmp/coop.npz
python tools/ab_bits.py cmp /tmp/nocoop.npz /tmp/coop.npz
for c in 2 4 8; do BK_COOP=$c timeout -k 10 200 python tools/coop_probe.py; done
