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
for world, threads in ((8, 4), (4, 4)):
    for leaves in (8, 12, 16):
        for cap in (128, 188, 252, 380, 508, 764, 0):
            for et in (2, 4):
                best = None
                for _ in range(3):
                    local, total = selfplay.self_play(ev, n_games=512, rollouts=400, rank=0, world=world, cap=8192, threads=threads, leaves=leaves, task_cap=cap, eager_top=et)
                    best = local["seconds"] if best is None else min(best, local["seconds"])
                print(f"world {world} leaves {leaves} task_cap {cap} eager_top {et}: {best:.4f} s steps {local['steps']} rows/step {local['rows_sent']/max(1,local['steps']):.0f} vevals {total['value_evals']:.0f}", flush=True)
