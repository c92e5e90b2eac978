#!/bin/bash
# the N-rank bench entry, rehearsed on one card over gloo (own launcher and torch.distributed.run form), three times each
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/probe12
export BK_BENCH_BACKEND=gloo BK_BENCH_DEVICE=0
for i in 1 2 3; do
  timeout -k 10 280 python3 bench.py --gpus 2 --steps 5 --warmup 2 --sustain 0 --no-f16x2 > gpurun_out/probe12/own_$i.json 2> gpurun_out/probe12/own_$i.err || { echo "own launcher run $i failed rc=$?"; tail -20 gpurun_out/probe12/own_$i.err; exit 1; }
  python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('own', d['n_gpus'], d['collective_ranks_seen'], round(d['value']), round(d['selfplay']['games_per_min']), d['launched_by'])" gpurun_out/probe12/own_$i.json || exit 1
  timeout -k 10 280 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port $((29600 + i)) bench.py --gpus 3 --steps 5 --warmup 2 --sustain 0 --no-f16x2 > gpurun_out/probe12/tr_$i.json 2> gpurun_out/probe12/tr_$i.err || { echo "torchrun run $i failed rc=$?"; tail -20 gpurun_out/probe12/tr_$i.err; exit 1; }
  python3 -c "import json,sys; d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{')][-1]; print('torchrun', d['n_gpus'], d['collective_ranks_seen'], round(d['value']), round(d['selfplay']['games_per_min']), d['launched_by'])" gpurun_out/probe12/tr_$i.json || exit 1
done
