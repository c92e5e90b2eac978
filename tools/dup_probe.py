"""How many of a self-play generation's leaf requests are repeats of a position already requested (by any game)?"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from bokego_amd import selfplay
from bokego_amd.bkw import load_bkw
from bokego_amd.engine import LeafEngine
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
eng = LeafEngine(load_bkw(os.path.join(g, "policy_19.bkw")), load_bkw(os.path.join(g, "value_synth.bkw")), max_batch=8192)
ev = selfplay.EngineEvaluator(eng)
prm = selfplay.search_params(rollouts=400, expand_thresh=100, noise_weight=0.25, sample_plies=8, max_turns=80, prune=1)
pool = selfplay.GamePool([20260 + x for x in range(256)], prm, cap=8192)
seen = set(); tot = dup_hist = dup_batch = 0; step = 0
by_step = []
while True:
    recs, npol = pool.collect_positions()
    if len(recs) == 0: break
    h = recs[:, 184:192].copy().view(np.uint64).reshape(-1)
    # value requests depend on the position only (hash covers stones, ko, side to move); last_move differs -> planes differ!
    lm = recs[:, 166:168].copy().view(np.int16).reshape(-1).astype(np.int64)
    key = [(int(a), int(b)) for a, b in zip(h, lm)]
    u = set(key)
    dup_batch += len(key) - len(u)
    dup_hist += sum(1 for k in u if k in seen)
    seen |= u
    tot += len(key); step += 1
    if step in (1, 2, 5, 10, 20, 50, 100, 200): by_step.append((step, tot, dup_batch, dup_hist))
    out = eng.wait(eng.submit_positions(recs, logits=False, probs=npol > 0, value=True, n_policy=npol))
    pool.deliver(selfplay.normalise_like_categorical(out["probs"]) if npol else np.zeros((0, 81), np.float32), out["value"])
print("requests", tot, "in-batch duplicates", dup_batch, "repeats of earlier batches", dup_hist, "-> saved fraction", (dup_batch + dup_hist) / tot)
print(by_step)
