"""Soak of the self-play driver: the 512-game generation of BASELINE configs[3] over and over in one process, alternating
precisions and host thread counts; every generation must reproduce the first one's games move for move.
    python tools/selfplay_soak.py [generations=60]"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402,F401
from bokego_amd import selfplay  # noqa: E402
from bokego_amd.bkw import load_bkw  # noqa: E402
from bokego_amd.engine import LeafEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
pw, vw = load_bkw(os.path.join(g, "policy_19.bkw")), load_bkw(os.path.join(g, "value_synth.bkw"))
engs = {p: LeafEngine(pw, vw, device_id=0, max_batch=8192, precision=p) for p in ("f32", "f16x2")}
first, bad, t0 = {}, 0, time.perf_counter()
for i in range(n):
    prec = ("f32", "f16x2")[i % 2]
    threads = (12, 4, 8, 2, 6)[i % 5]
    pools = (2, 3, 2, 1)[i % 4]
    local, total = selfplay.self_play(selfplay.EngineEvaluator(engs[prec]), n_games=512, rollouts=400, cap=8192, threads=threads, n_pools=pools)
    h = hashlib.sha256(repr(sorted((k, v["moves"], v["score"]) for k, v in local["games"].items())).encode()).hexdigest()[:16]
    if first.setdefault(prec, h) != h:
        bad += 1
        print(f"generation {i} ({prec}, {threads} threads, {pools} pools): games differ ({h} vs {first[prec]})", flush=True)
    if i % 10 == 9:
        print(f"{i + 1} generations, {bad} differing, {time.perf_counter() - t0:.0f} s; last: {prec} {local['seconds']:.2f} s", flush=True)
print(f"{n} generations of 512 games: {bad} differing; games digests {first}")
sys.exit(1 if bad else 0)
