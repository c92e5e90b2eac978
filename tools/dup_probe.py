"""How many of a self-play generation's network tasks are repeats of a position another game has already asked for?  (VERDICT r3 item 7:
the reference keeps ONE memo for all trees -- mcts.py:41-44, class-level _val_cache / _dist_cache --; here every game has its own.)
The generation of bench.py's configs[3] leg: 512 games x 400 rollouts/move in two pools of 256, an expansion evaluates its eager_top
best-prior children (fp32: 4), batches held to whole rounds of workgroups.  A row's identity = (position hash, last move, policy|value):
the planes depend on the last move, and a policy row runs both nets (2 tasks), a value row one.
    python tools/dup_probe.py [games=512] [eager_top=4]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from bokego_amd import selfplay
from bokego_amd.bkw import load_bkw
from bokego_amd.engine import LeafEngine
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
n_games = int(sys.argv[1]) if len(sys.argv) > 1 else 512
et = int(sys.argv[2]) if len(sys.argv) > 2 else 4
eng = LeafEngine(load_bkw(os.path.join(g, "policy_19.bkw")), load_bkw(os.path.join(g, "value_synth.bkw")), max_batch=8192)
prm = selfplay.search_params(rollouts=400, expand_thresh=100, noise_weight=0.25, sample_plies=8, max_turns=80, prune=1, eager_top=et)
parts = [list(range(n_games))[i::2] for i in range(2)]
pools = [selfplay.GamePool([20260 + x for x in part], prm, cap=8192) for part in parts]
cap = 768 * max(1, round(3.0 * len(parts[0]) / 768)) - 4
for p in pools:
    p.set_task_cap(cap)
seen = set()
T = dict(tasks=0, in_batch=0, history=0, rows=0, steps=0)
by_ply = {}
live = True
while live:
    live = False
    for pool in pools:
        recs, npol = pool.collect_positions()
        if len(recs) == 0:
            continue
        live = True
        h = recs[:, 184:192].copy().view(np.uint64).reshape(-1)
        lm = recs[:, 166:168].copy().view(np.int16).reshape(-1).astype(np.int64)
        stones = (recs[:, :81] != 0).sum(1)
        first = set()
        for i in range(len(recs)):
            kind, w = (0, 2) if i < npol else (1, 1)
            # a value is also known once the position's policy row went out (that row runs both nets)
            k = (int(h[i]), int(lm[i]), kind)
            kp = (int(h[i]), int(lm[i]), 0)
            T["tasks"] += w
            b = by_ply.setdefault(int(stones[i]) // 8, [0, 0])
            b[0] += w
            if k in first or (kind == 1 and kp in first):
                T["in_batch"] += w; b[1] += w
            elif k in seen or (kind == 1 and kp in seen):
                T["history"] += w; b[1] += w
            first.add(k)
        seen |= first
        T["rows"] += len(recs); T["steps"] += 1
        out = eng.wait(eng.submit_positions(recs, logits=False, probs=npol > 0, value=True, n_policy=npol))
        pool.deliver(selfplay.normalise_like_categorical(out["probs"]) if npol else np.zeros((0, 81), np.float32), out["value"])
print(f"{n_games} games, eager_top {et}: {T['steps']} steps, {T['rows']} rows = {T['tasks']} network tasks; duplicates inside a batch "
      f"{T['in_batch']} ({100 * T['in_batch'] / T['tasks']:.2f} %), repeats of an earlier request of any game {T['history']} "
      f"({100 * T['history'] / T['tasks']:.2f} %) -> a generation-wide memo + in-batch de-duplication would save "
      f"{100 * (T['in_batch'] + T['history']) / T['tasks']:.2f} % of the tasks")
print("by stones on the board (x8):", {k: f"{100 * v[1] / max(1, v[0]):.1f}% of {v[0]}" for k, v in sorted(by_ply.items())})
