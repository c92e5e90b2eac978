"""One rank's share of configs[3] at 8 ranks (64 games) and at 4 (128): pools x batch limit (bk_pool_set_task_cap) sweep.
    python tools/pools_probe.py"""
import os, sys
sys.path.insert(0, os.getcwd())
from bokego_amd import selfplay
from bokego_amd.bkw import load_bkw
from bokego_amd.engine import LeafEngine
g = "tests/golden"
eng = LeafEngine(load_bkw(f"{g}/policy_19.bkw"), load_bkw(f"{g}/value_synth.bkw"), max_batch=8192)
ev = selfplay.EngineEvaluator(eng)
selfplay.self_play(ev, n_games=64, rollouts=50, cap=8192)
for games in (64, 128):
    for pools in (1, 2, 3, 4):
        for tc in (None, 64, 85, 128, 170):
            if tc is not None and pools == 1:
                continue
            best = None
            for _ in range(3):
                local, total = selfplay.self_play(ev, n_games=games, rollouts=400, cap=8192, threads=4, n_pools=pools, task_cap=tc)
                best = local["seconds"] if best is None else min(best, local["seconds"])
            print(f"{games} games, pools {pools}, task_cap {'default' if tc is None else tc}: {best:.3f} s = {games / best * 60:,.0f} games/min, "
                  f"steps {local['steps']}, mean rows {local['rows_sent'] / max(1, local['steps']):.0f}", flush=True)
