"""One rank's share of the 512-game configs[3] job at 1 / 2 / 4 / 8 ranks (512 / 256 / 128 / 64 games on ONE card), fp32, the host threads
bench.py plans per rank; games/min of the node if every rank did the same.  Strong scaling by construction (the job is 512 games).
    python tools/shard_share.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from bokego_amd import selfplay
from bokego_amd.bkw import load_bkw
from bokego_amd.engine import LeafEngine
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
eng = LeafEngine(load_bkw(os.path.join(g, "policy_19.bkw")), load_bkw(os.path.join(g, "value_synth.bkw")), max_batch=8192)
ev = selfplay.EngineEvaluator(eng)
selfplay.self_play(ev, n_games=64, rollouts=50, cap=8192)
first = None
for world, threads in ((1, 12), (2, 8), (4, 4), (8, 4), (8, 2), (8, 1)):   # (8, 1): what a node that owns only 16 CPUs could give a rank
    n = 512 // world
    best = None
    for _ in range(3):
        local, total = selfplay.self_play(ev, n_games=512, rollouts=400, rank=0, world=world, cap=8192, threads=threads)
        best = local["seconds"] if best is None else min(best, local["seconds"])
    first = best if first is None else first
    print(f"world {world}: {n} games on this rank, {threads} host threads: {best:.3f} s -> {512 / best * 60:,.0f} games/min for the node "
          f"({100 * first / (world * best):.0f} % of linear)", flush=True)
