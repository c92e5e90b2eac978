# the last A/B run around the small-batch path on the GPU box (edit freely)
cd /root/repo
export PYTHONPATH=/root/repo
run() { echo "== $*"; env "$@" timeout -k 10 200 python tools/genmove_probe.py 40 $PREC 2>&1 | grep -v amdgpu | head -4; }
PREC=f32
run BK_SPECULATE=0
for s in 40 50 60 70; do run BK_SPECULATE=$s BK_SPECULATE_ROWS=64 BK_SPECULATE_ASYNC=1; done
run BK_SPECULATE=50 BK_SPECULATE_ROWS=128 BK_SPECULATE_ASYNC=1
PREC=f16x2
run BK_SPECULATE=50 BK_SPECULATE_ROWS=256 BK_SPECULATE_ASYNC=0
run BK_SPECULATE=50 BK_SPECULATE_ROWS=128 BK_SPECULATE_ASYNC=1
run BK_SPECULATE=50 BK_SPECULATE_ROWS=256 BK_SPECULATE_ASYNC=1
