"""eager_top (children evaluated at an expansion) x batch limit at the shares of configs[3] a rank gets at 1 / 2 / 4 / 8 ranks.
    python tools/eager_probe.py"""
import os, sys
sys.path.insert(0, os.getcwd())
from bokego_amd import selfplay
from bokego_amd.bkw import load_bkw
from bokego_amd.engine import LeafEngine
g = "tests/golden"
eng = LeafEngine(load_bkw(f"{g}/policy_19.bkw"), load_bkw(f"{g}/value_synth.bkw"), max_batch=8192)
ev = selfplay.EngineEvaluator(eng)
selfplay.self_play(ev, n_games=64, rollouts=50, cap=8192)
for games, threads in ((512, 12), (256, 8), (128, 4), (64, 4)):
    for et, tc in ((4, None), (2, None), (2, 508), (2, 252), (3, None), (3, 508)):
        if tc is not None and tc > 1.2 * games:
            continue
        best = None
        for _ in range(3):
            local, total = selfplay.self_play(ev, n_games=games, rollouts=400, cap=8192, threads=threads, eager_top=et, task_cap=tc)
            best = local["seconds"] if best is None else min(best, local["seconds"])
        print(f"{games} games, eager_top {et}, task_cap {'default' if tc is None else tc}: {best:.3f} s = {games / best * 60:,.0f} games/min, "
              f"steps {local['steps']}, value evals {int(total['value_evals'])}, plies {int(total['plies'])}", flush=True)
