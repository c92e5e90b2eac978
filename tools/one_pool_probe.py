"""A rank's small share of configs[3] with ONE pool (no host / GPU overlap, but requests twice as large) against the default two."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from bokego_amd import selfplay
from bokego_amd.bkw import load_bkw
from bokego_amd.engine import LeafEngine
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
eng = LeafEngine(load_bkw(g + "/policy_19.bkw"), load_bkw(g + "/value_synth.bkw"), max_batch=8192)
ev = selfplay.EngineEvaluator(eng)
selfplay.self_play(ev, n_games=64, rollouts=50, cap=8192)
ref = {}
for world, threads in ((8, 4), (4, 4)):
    for pools, tc, spec in ((2, None, None), (1, None, None), (1, 188, 0), (1, 252, 0), (1, 188, 70), (1, 252, 70), (1, 380, 0)):
        best = None
        for _ in range(3):
            local, total = selfplay.self_play(ev, n_games=512, rollouts=400, rank=0, world=world, cap=8192, threads=threads, n_pools=pools, task_cap=tc, speculate=spec)
            best = local["seconds"] if best is None else min(best, local["seconds"])
            assert ref.setdefault(world, local["games"]) == local["games"]
        print(f"world {world}: pools {pools} task_cap {tc} speculate {spec}: {best:.4f} s steps {local['steps']} rows/step {local['rows_sent']/max(1,local['steps']):.0f} caps {local['task_caps']}", flush=True)
