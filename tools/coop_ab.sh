# the last A/B run around the cooperative launch on the GPU box (edit freely)
# self-play generations of 8 .. 128 games (a rank's share of a 512-game job on 8 GPUs is 64): cooperative launch on / off
cd /root/repo
export PYTHONPATH=/root/repo
for g in 8 16 64 128; do
  for c in 0 on 0 on 0 on; do
    if [ $c = 0 ]; then export BK_COOP=0; else unset BK_COOP; fi
    s=$(timeout -k 10 200 python -m bokego_amd.selfplay --games $g --rollouts 400 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['seconds'])")
    echo "games $g coop=$c seconds $s"
  done
done
