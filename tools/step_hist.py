import os, sys
sys.path.insert(0, os.getcwd())
from collections import Counter
from bokego_amd import selfplay
from bokego_amd.bkw import load_bkw
from bokego_amd.engine import LeafEngine
g = "tests/golden"
eng = LeafEngine(load_bkw(f"{g}/policy_19.bkw"), load_bkw(f"{g}/value_synth.bkw"), max_batch=8192)
ev = selfplay.EngineEvaluator(eng)
hist = Counter()
orig = ev.submit
def sub(feats, npol):
    t = len(feats) + npol
    hist["<=128" if t <= 128 else "129-192" if t <= 192 else "193-256" if t <= 256 else "257-384" if t <= 384 else "385-512" if t <= 512 else "513-768" if t <= 768 else ">768"] += 1
    return orig(feats, npol)
ev.submit = sub
selfplay.self_play(ev, n_games=64, rollouts=50, cap=8192)
for games in (512, 256, 128):
    hist.clear()
    s0 = eng.stats()
    local, total = selfplay.self_play(ev, n_games=games, rollouts=400, cap=8192)
    s1 = eng.stats()
    print(games, "games:", f"{local['seconds']:.3f} s", dict(hist), "coop launches", s1["coop_launches"] - s0["coop_launches"], "fallbacks", s1["coop_fallbacks"] - s0["coop_fallbacks"], "batches", s1["batches"] - s0["batches"], flush=True)
