"""fp32 self-play of bench.py's configs[3] leg with and without in-batch de-duplication, alternating on one box.
    python tools/dedup_ab.py [games=512] [reps=3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
from bokego_amd import selfplay
from bokego_amd.bkw import load_bkw
from bokego_amd.engine import LeafEngine
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
prec = sys.argv[3] if len(sys.argv) > 3 else "f32"
eng = LeafEngine(load_bkw(os.path.join(g, "policy_19.bkw")), load_bkw(os.path.join(g, "value_synth.bkw")), max_batch=8192, precision=prec)
ev = selfplay.EngineEvaluator(eng)
threads = selfplay.default_threads(n // 2)
selfplay.self_play(ev, n_games=min(64, n), rollouts=50, cap=8192, threads=threads)
ref = None
for r in range(reps):
    for dd in (False, True):
        local, total = selfplay.self_play(ev, n_games=n, rollouts=400, cap=8192, threads=threads, dedup=dd)
        moves = {k: v["moves"] for k, v in local["games"].items()}
        if ref is None:
            ref = moves
        assert moves == ref, "de-duplication changed a game"
        print(f"{prec} {n} games dedup={int(dd)}: {local['seconds']:.3f} s = {n / local['seconds'] * 60:,.0f} games/min, steps {local['steps']}, "
              f"rows requested {local['rows_requested']} sent {local['rows_sent']}, value evals {int(total['value_evals'])}", flush=True)
