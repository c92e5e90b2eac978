"""Where a self-play generation's wall time goes on the host: blocked on the GPU (wait), tree work (deliver +
the advance half of collect), feature encoding, submit.  usage (GPU box): python tools/selfplay_breakdown.py [pools] [host|gpu (where the planes are encoded)] [games] [eager_top] [threads]"""
import os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from bokego_amd import selfplay
from bokego_amd.bkw import load_bkw
from bokego_amd.engine import LeafEngine

n_pools = int(sys.argv[1]) if len(sys.argv) > 1 else 3
gpu_encode = (sys.argv[2] if len(sys.argv) > 2 else "gpu") == "gpu"
n_games = int(sys.argv[3]) if len(sys.argv) > 3 else 512
eager_top = int(sys.argv[4]) if len(sys.argv) > 4 else 4       # children evaluated per expansion (0: all)
threads = int(sys.argv[5]) if len(sys.argv) > 5 else None
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
eng = LeafEngine(load_bkw(os.path.join(g, "policy_19.bkw")), load_bkw(os.path.join(g, "value_synth.bkw")), max_batch=4096)
ev = selfplay.EngineEvaluator(eng, gpu_encode=gpu_encode)
prm = selfplay.search_params(rollouts=400, expand_thresh=100, noise_weight=0.25, sample_plies=8, max_turns=80, prune=1, eager_top=eager_top)
gids = list(range(n_games))
pools = [selfplay.GamePool([20260 + x for x in gids[i::n_pools]], prm, cap=4096, threads=threads) for i in range(n_pools)]
T = dict(wait=0.0, deliver=0.0, collect=0.0, submit=0.0)
inflight = [None] * n_pools
live = [True] * n_pools
steps = pos = 0
t_all = time.perf_counter()
while any(live) or any(h is not None for h in inflight):
    for i, pool in enumerate(pools):
        if inflight[i] is not None:
            t = time.perf_counter(); probs, values = ev.finish(inflight[i]); T["wait"] += time.perf_counter() - t
            t = time.perf_counter(); pool.deliver(probs, values); T["deliver"] += time.perf_counter() - t
            inflight[i] = None
        if live[i]:
            t = time.perf_counter(); feats, npol = (pool.collect_positions() if gpu_encode else pool.collect()); T["collect"] += time.perf_counter() - t
            if len(feats) == 0:
                live[i] = False
            else:
                t = time.perf_counter(); inflight[i] = ev.submit(feats, npol); T["submit"] += time.perf_counter() - t
                steps += 1; pos += len(feats)
tot = time.perf_counter() - t_all
print(f"games {n_games}, pools {n_pools}, planes encoded on the {'GPU' if gpu_encode else 'host'}: {tot:.3f} s, {steps} steps, mean batch {pos / steps:.0f}, {n_games / tot * 60:.0f} games/min")
for k, v in T.items():
    print(f"  {k:8s} {v:.3f} s  {100 * v / tot:5.1f} %  {v / steps * 1e3:.3f} ms/step")
