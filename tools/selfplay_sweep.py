"""In-process sweep of self-play settings (fresh processes are noisy under a CPU-time quota: tools/r03_probe17.sh): every setting
several times, interleaved, best and median seconds of the 512-game generation.
    python tools/selfplay_sweep.py f16x2 "eager_top=4,6" "task_cap=0,764,1532" [reps=3]"""
import itertools
import os
import statistics
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402,F401
from bokego_amd import selfplay  # noqa: E402
from bokego_amd.bkw import load_bkw  # noqa: E402
from bokego_amd.engine import LeafEngine  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x2"
axes = {}
reps = 3
for a in sys.argv[2:]:
    if "=" in a:
        k, v = a.split("=")
        axes[k] = [int(x) for x in v.split(",")]
    else:
        reps = int(a)
g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
eng = LeafEngine(load_bkw(os.path.join(g, "policy_19.bkw")), load_bkw(os.path.join(g, "value_synth.bkw")), device_id=0, max_batch=8192, precision=prec)
ev = selfplay.EngineEvaluator(eng)
selfplay.self_play(ev, n_games=64, rollouts=100)
combos = [dict(zip(axes, v)) for v in itertools.product(*axes.values())] or [{}]
times = {i: [] for i in range(len(combos))}
for _ in range(reps):
    for i, kw in enumerate(combos):
        local, total = selfplay.self_play(ev, n_games=512, rollouts=400, cap=8192, **kw)
        times[i].append(local["seconds"])
        assert total["black_wins"] == 235 and total["plies"] == 41464, (kw, total["black_wins"], total["plies"])
for i, kw in enumerate(combos):
    t = times[i]
    print(f"{prec} {kw}: best {min(t):.3f} s  median {statistics.median(t):.3f} s  = {512 / statistics.median(t) * 60:,.0f} games/min  all {[round(x, 3) for x in t]}")
