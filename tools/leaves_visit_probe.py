"""The multi-leaf mode's two kinds of virtual loss by share: a waiting rollout as a lost visit (default) or as a visit only (leaves_visit_only)."""
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
for world, threads in ((8, 4), (4, 4), (1, 12)):
    for leaves in (1, 4, 8, 16):
        for vo in ((0,) if leaves == 1 else (0, 1)):
            best = None
            for _ in range(3):
                local, total = selfplay.self_play(ev, n_games=512, rollouts=400, rank=0, world=world, cap=8192, threads=threads, leaves=leaves, leaves_visit_only=vo)
                best = local["seconds"] if best is None else min(best, local["seconds"])
            print(f"world {world}: {512 // world} games, leaves {leaves}, visit_only {vo}: {best:.3f} s; steps {local['steps']}, rows/step {local['rows_sent'] / max(1, local['steps']):.0f}, "
                  f"value evals {total['value_evals']:.0f}", flush=True)
