"""Unequal pools for small shards: a rank's 64 / 128 games split so that each pool's requests land on the cheap side of the fp32
launch forms' size steps (kernel time by tasks: <= 64: 104 us, <= 80: 135, <= 128: 174, <= 192: 249, <= 256: 298, <= 384: 422).
Two equal pools of 32 games ask for ~94 tasks each -- the dear end of the 81..128 range; 43 + 21 games ask for ~126 + ~62.
    python tools/pool_split_probe.py"""
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
cases = {
    64: [((32, 32), None), ((32, 32), (128, 128)), ((43, 21), (128, 64)), ((42, 22), (128, 64)), ((40, 24), (128, 64)), ((38, 26), (128, 80)),
         ((44, 20), (128, 64)), ((43, 21), None), ((40, 24), None), ((22, 21, 21), (64, 64, 64)), ((43, 21), (124, 62)), ((41, 23), (126, 64))],
    128: [((64, 64), None), ((86, 42), (252, 128)), ((84, 44), (252, 128)), ((80, 48), (252, 128)), ((86, 42), None), ((64, 43, 21), (188, 128, 64)),
          ((43, 43, 42), (128, 128, 128)), ((88, 40), (252, 124))],
}
for games, lst in cases.items():
    world = 512 // games
    for sizes, caps in lst:
        best = None
        for _ in range(3):
            local, total = selfplay.self_play(ev, n_games=512, rollouts=400, rank=0, world=world, cap=8192, threads=4, pool_sizes=sizes, task_cap=caps)
            best = local["seconds"] if best is None else min(best, local["seconds"])
            assert ref.setdefault(games, local["games"]) == local["games"]
        print(f"{games} games, pools {sizes}, task caps {caps or 'default'}: {best:.3f} s -> {512 / best * 60:,.0f} games/min for the node; "
              f"steps {local['steps']}, mean rows {local['rows_sent'] / max(1, local['steps']):.0f}", flush=True)
