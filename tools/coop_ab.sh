# the last A/B run around the cooperative launch on the GPU box (edit freely)
cd /root/repo
export PYTHONPATH=/root/repo
for g in 64; do
  for p in 1 2 3 4 1 2 3 4; do
    s=$(timeout -k 10 200 python -m bokego_amd.selfplay --games $g --rollouts 400 --pools $p 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['seconds'], d.get('mean_batch'), d.get('batches'))")
    echo "games $g pools $p seconds $s"
  done
done
