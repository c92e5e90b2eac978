"""finer sweep around tools/spec_probe.py's best point for a 64-game shard (speculate x task cap x rows per game x pools)"""
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
ref = {}
rows_out = []
for world in (8,):
    for spec in (0, 60, 65, 70, 75, 80, 90):
        for rows in (4, 8, 12):
            for tc in (112, 120, 124, 128):
                for et in (4,):
                    if spec == 0 and (rows != 8 or tc != 128):
                        continue
                    best = None
                    for _ in range(3):
                        local, total = selfplay.self_play(ev, n_games=512, rollouts=400, rank=0, world=world, cap=8192, threads=4, speculate=spec,
                                                          speculate_rows=rows, task_cap=tc, eager_top=et)
                        best = local["seconds"] if best is None else min(best, local["seconds"])
                        assert ref.setdefault(world, local["games"]) == local["games"]
                    rows_out.append((best, f"world {world}: speculate {spec}, rows/game {rows}, task_cap {tc}, eager_top {et}: {best:.4f} s; steps {local['steps']}, "
                                           f"mean rows {local['rows_sent'] / max(1, local['steps']):.0f}, value evals {total['value_evals']:.0f}"))
                    print(rows_out[-1][1], flush=True)
print("--- best five")
for b, line in sorted(rows_out)[:5]:
    print(line)
