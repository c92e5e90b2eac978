"""Evaluation ahead of expansion (bk_search_params.speculate) in the lock-step pools of a small shard: a leaf that reaches N
visits sends its policy row -- and, once its priors are back, its best-prior children -- with requests that go out anyway, so
that its expansion (at 100 visits) needs no round trip of its own.  A 32-game pool's request (~94 tasks) runs as the
2-CUs-per-board launch, whose time is flat from 81 to 128 tasks: the extra rows are free, fewer steps are not.
    python tools/spec_probe.py [--fine]      (--fine: the 64-game share only, speculate x rows per game x task cap)"""
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
if "--fine" in sys.argv:
    out = []
    for spec in (0, 60, 70, 80, 90):
        for rows in (4, 8):
            for tc in (88, 92, 96, 100, 112, 128, 0):
                if spec == 0 and rows != 8:
                    continue
                best = None
                for _ in range(3):
                    local, total = selfplay.self_play(ev, n_games=512, rollouts=400, rank=0, world=8, cap=8192, threads=4, speculate=spec,
                                                      speculate_rows=rows, task_cap=tc, eager_top=4)
                    best = local["seconds"] if best is None else min(best, local["seconds"])
                    assert ref.setdefault(8, local["games"]) == local["games"]
                out.append((best, f"world 8: speculate {spec}, rows/game {rows}, task_cap {tc}, eager_top 4: {best:.4f} s; steps {local['steps']}, "
                                  f"mean rows {local['rows_sent'] / max(1, local['steps']):.0f}, value evals {total['value_evals']:.0f}"))
                print(out[-1][1], flush=True)
    print("--- best five")
    for _, line in sorted(out)[:5]:
        print(line)
    sys.exit(0)
for world, threads in ((8, 4), (4, 4), (2, 8), (1, 12)):
    for spec, rows, tc in ((0, 8, None), (50, 8, None), (70, 8, None), (30, 8, None), (50, 16, None), (50, 8, 128), (70, 8, 128), (85, 8, None)):
        if world <= 2 and (rows != 8 or tc is not None or spec == 30):
            continue
        best = None
        for _ in range(3):
            local, total = selfplay.self_play(ev, n_games=512, rollouts=400, rank=0, world=world, cap=8192, threads=threads, speculate=spec,
                                              speculate_rows=rows, task_cap=tc)
            best = local["seconds"] if best is None else min(best, local["seconds"])
            assert ref.setdefault(world, local["games"]) == local["games"]          # the same games: evaluating ahead changes no search
        print(f"world {world}: {512 // world} games, speculate {spec}, rows/game {rows}, task_cap {tc or 'default'}: {best:.3f} s -> "
              f"{512 / best * 60:,.0f} games/min for the node; steps {local['steps']}, mean rows {local['rows_sent'] / max(1, local['steps']):.0f}, "
              f"value evals {total['value_evals']:.0f}", flush=True)
