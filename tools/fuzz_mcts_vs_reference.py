#!/usr/bin/env python3
"""The REFERENCE's MCTS (bokego/mcts.py) and this package's Python tree (bokego_amd/mcts.py) side by side on the reference's own
torch networks, under random keyword arguments (expand_thresh, branch_num, exploration_weight) and random sequences of rollouts,
choose() and outside moves: the root and N / V of its children must be equal after every step.  A check to run where the
reference checkout is (BOKEGO_REFERENCE); nothing of it travels.
    python tools/fuzz_mcts_vs_reference.py [seed] [seconds]"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import build_nets, go as rgo, mcts as rmcts, torch  # noqa: E402  (seeds `random` before importing the reference)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bokego_amd import mcts as M  # noqa: E402

pi, v = build_nets()[:2]
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 120)
cases = steps = 0
while time.time() < t_end:
    kw = dict(expand_thresh=rng.choice([1, 2, 3, 5, 8, 20]), exploration_weight=rng.choice([4.0, 4.0, 1.5, 8.0]))
    bn = rng.choice([None, None, 5, 12, 30])
    if bn:
        kw["branch_num"] = bn
    for c in (rmcts.MCTS._val_cache, rmcts.MCTS._dist_cache, rmcts.MCTS._fts_cache):
        c.clear()
    torch.manual_seed(cases)
    ref = rmcts.MCTS(rmcts.Go_MCTS(), pi, v, no_sim=True, **kw)
    torch.manual_seed(cases)
    our = M.MCTS(M.Go_MCTS(), pi, v, no_sim=True, **kw)
    for step in range(rng.randint(2, 8)):
        op = rng.random()
        if op < 0.6:
            n = rng.randint(1, 60)
            ref.rollout(n); our.rollout(n)
        elif op < 0.85:
            kids = ref.children.get(ref.root)
            if not kids:
                break
            top = sorted((int(ref.N[c]) for c in kids), reverse=True)
            if top[0] == 0 or (len(top) > 1 and top[0] == top[1]):
                continue         # a tie at the top: the reference takes the first of a SET of hashed nodes, this build the lowest move
            a, b = ref.choose(), our.choose()
            assert a.last_move == b.last_move, ("choose", kw, step, a.last_move, b.last_move)
        else:
            legal = sorted(ref.root.get_legal_moves())
            if not legal:
                break
            mv = rng.choice(legal)
            ref.set_root(ref.root.make_move(mv)); our.set_root(our.root.make_move(mv))
        assert our.root.board == ref.root.board and our.root.ko == ref.root.ko and our.root.turn == ref.root.turn, ("root", kw, step)
        if ref.root._terminal:
            break
        want = {int(c.last_move): (int(ref.N[c]), float(ref.V[c])) for c in ref.children[ref.root]}
        got = {int(c.mv): (int(our.N[c]), float(our.V[c])) for c in our.children[our.root]}
        assert set(got) == set(want), ("children", kw, step, sorted(set(got) ^ set(want)))
        bad = [(k, got[k], want[k]) for k in want if got[k][0] != want[k][0] or abs(got[k][1] - want[k][1]) > 1e-5 * max(1, got[k][0])]
        assert not bad, ("stats", kw, step, bad[:4])
        steps += 1
    cases += 1
print(f"{cases} searches, {steps} compared steps: equal")
