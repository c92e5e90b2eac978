# the last A/B run around the cooperative launch on the GPU box (edit freely)
set -e
cd /root/repo
export PYTHONPATH=/root/repo
BK_COOP=0 timeout -k 10 200 python tools/ab_bits.py dump /tmp/nocoop.npz
timeout -k 10 200 python tools/ab_bits.py dump /tmp/coop.npz
python tools/ab_bits.py cmp /tmp/nocoop.npz /tmp/coop.npz
BK_LIB_PATH=/root/repo/bokego_amd/libbokego_amd_old.so BK_LIB_ANY_ABI=1 timeout -k 10 200 python tools/ab_bits.py dump /tmp/old.npz
python tools/ab_bits.py cmp /tmp/old.npz /tmp/coop.npz
timeout -k 10 200 python tools/coop_probe.py
BK_LIB_PATH=bokego_amd/libbokego_amd_diag.so timeout -k 10 200 python tools/stamp_coop.py 62
