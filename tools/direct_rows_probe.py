import os, sys
sys.path.insert(0, "/root/repo")
import torch
from bokego_amd import selfplay
from bokego_amd.bkw import load_bkw
from bokego_amd.engine import LeafEngine
g = "/root/repo/tests/golden"
eng = LeafEngine(load_bkw(g + "/policy_19.bkw"), load_bkw(g + "/value_synth.bkw"), max_batch=8192)
ev = selfplay.EngineEvaluator(eng)
selfplay.self_play(ev, n_games=64, rollouts=50, cap=8192)
ref = {}
for rep in range(2):
    for dr in (256, 512, 1024, 4096):
        eng.set_option("direct_rows", dr)
        for world, threads in ((1, 12), (2, 8)):
            best = None
            for _ in range(3):
                local, total = selfplay.self_play(ev, n_games=512, rollouts=400, rank=0, world=world, cap=8192, threads=threads)
                best = local["seconds"] if best is None else min(best, local["seconds"])
                assert ref.setdefault(world, local["games"]) == local["games"]
            print(f"direct_rows {dr}: world {world}: {best:.4f} s steps {local['steps']}", flush=True)
