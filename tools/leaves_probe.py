"""The opt-in multi-leaf mode (bk_search_params.leaves; SURVEY 7.6, VERDICT r5 next #7 -- outside SURVEY 8, no parity claim):
(1) a rank's share of configs[3] at 8 / 4 / 1 ranks with 1 (the reference's search), 2, 4, 8 leaves per tree and step: seconds (best of
three), steps, rows per step, evaluations; (2) a 100-game match of the multi-leaf search against the one-leaf search through
bokego_amd.match (same nets, same rollouts per move, colours alternating, seeded 4-ply openings): win rate and ms/move.
    python tools/leaves_probe.py [--quick] [--match-only | --no-match]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from bokego_amd import match, nnet, selfplay  # noqa: E402
from bokego_amd.bkw import load_bkw  # noqa: E402
from bokego_amd.engine import LeafEngine  # noqa: E402

g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
quick = "--quick" in sys.argv
if "--match-only" not in sys.argv:
    eng = LeafEngine(load_bkw(os.path.join(g, "policy_19.bkw")), load_bkw(os.path.join(g, "value_synth.bkw")), max_batch=8192)
    ev = selfplay.EngineEvaluator(eng)
    selfplay.self_play(ev, n_games=64, rollouts=50, cap=8192)
    for world, threads in ((8, 4), (4, 4), (1, 12)) if not quick else ((8, 4),):
        games = {}
        for leaves in (1, 2, 4, 8, 16, 32) if "--ab" not in sys.argv else (1, 8):
            best = None
            for _ in range(3):
                local, total = selfplay.self_play(ev, n_games=512, rollouts=400, rank=0, world=world, cap=8192, threads=threads, leaves=leaves)
                best = local["seconds"] if best is None else min(best, local["seconds"])
                assert games.setdefault(leaves, local["games"]) == local["games"]        # a pure function of the seeds, run after run
            i_val, i_pol = selfplay.STATS_FIELDS.index("value_evals"), selfplay.STATS_FIELDS.index("policy_evals")
            ve, pe = local["local_stats"][i_val], local["local_stats"][i_pol]
            tf = (pe * 133_413_888 + ve * 133_424_384) / best / 1e12
            print(f"world {world}: {512 // world} games, leaves {leaves}: {best:.3f} s -> {512 / best * 60:,.0f} games/min for the node; steps {local['steps']}, "
                  f"rows/step {local['rows_sent'] / max(1, local['steps']):.0f}, value evals {ve:.0f}, policy {pe:.0f}, {tf:.1f} TFLOP/s = {tf / 157.3:.3f} of peak, "
                  f"task cap {local['task_caps'][0]}, mean plies {total['plies'] / total['games']:.1f}, black wins {total['black_wins']:.0f}", flush=True)
    eng.close()
if "--no-match" not in sys.argv:
    from bokego_amd.gtp import NativeGTP
    from bokego_amd.mcts_native import Position
    pw, vw = load_bkw(os.path.join(g, "policy_19.bkw")), load_bkw(os.path.join(g, "value_synth.bkw"))
    pi, val = nnet.HipPolicyNet(pw), nnet.HipValueNet(vw)
    n_games = 20 if quick else 100
    for leaves, rollouts in ((8, 1600), (16, 1600), (8, 400), (16, 400), (32, 400)) if not quick else ((4, 400),):
        a = match.InProcessEngine(NativeGTP(Position(), pi, val, no_sim=True, time_lim=None, n_rollouts=rollouts, leaves=leaves), name=f"leaves{leaves}")
        b = match.InProcessEngine(NativeGTP(Position(), pi, val, no_sim=True, time_lim=None, n_rollouts=rollouts), name="one_leaf")
        res = match.play_match(a, b, n_games, 5.5, None, opening_plies=4, seed=60_000)
        res.pop("records")
        res.update(rollouts=rollouts, leaves=leaves, opening_plies=4)
        print(json.dumps(res), flush=True)
