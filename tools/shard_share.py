"""One rank's share of the 512-game configs[3] job at 1 / 2 / 4 / 8 ranks (512 / 256 / 128 / 64 games on ONE card), fp32, the host threads
bench.py plans per rank; games/min of the node if every rank did the same.  Strong scaling by construction (the job is 512 games).
Round 5: the pools' step loop in C (bk_pools_run) against the same loop in Python, and the number of pools, for every share.
    python tools/shard_share.py [--quick]"""
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
quick = "--quick" in sys.argv
ref_games = {}
first = {}
for world, threads in ((1, 12), (2, 8), (4, 4), (8, 4)) if not quick else ((8, 4), (4, 4)):
    n = 512 // world
    for loop in ("python", "C"):
        for pools, tc in ((2, None), (3, None), (4, None), (3, 64), (3, 85), (4, 64)) if n <= 128 else ((2, None), (3, None)):
            best = None
            for _ in range(3):
                local, total = selfplay.self_play(ev, n_games=512, rollouts=400, rank=0, world=world, cap=8192, threads=threads,
                                                  n_pools=pools, task_cap=tc, native_loop=loop == "C")
                best = local["seconds"] if best is None else min(best, local["seconds"])
                assert ref_games.setdefault(world, local["games"]) == local["games"]     # the same games, whatever the loop and the grouping
            key = (loop, pools, tc)
            if world == 1 or quick:
                first.setdefault(key, best)
            lin = first.get(key) or first.get(("C", 2, None)) or best
            print(f"world {world}: {n} games, {threads} host threads, loop {loop:6s}, pools {pools}, task_cap {'default' if tc is None else tc}: "
                  f"{best:.3f} s -> {512 / best * 60:,.0f} games/min for the node; steps {local['steps']}, "
                  f"mean rows {local['rows_sent'] / max(1, local['steps']):.0f}", flush=True)
