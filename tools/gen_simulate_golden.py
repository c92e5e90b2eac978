#!/usr/bin/env python3
"""tests/golden/simulate_playouts.json: what the REFERENCE does in simulation mode (MCTS(no_sim=False), boke.py --simulate,
mcts.py:147-148,195-206), as far as it gets.

A rollout there ends in a playout whose moves Go_MCTS.get_move samples from the policy (mcts.py:348-360): a sampled move that
is illegal or fills the mover's own one-point eye is zeroed in the node's (cached) distribution and another one drawn.  The
"pass as a last resort" branch needs 81 rejections of moves with a non-zero probability and is unreachable: once every
acceptable move is used up -- on a 9x9 board some turns before MAX_TURNS = 80 -- the distribution is all zero and
torch.multinomial raises "invalid multinomial distribution".  With the published weights searches in this mode end that way
inside their first rollouts (recorded below: 10 seeds per setting, 40 rollouts asked for each), so there is no search trace
to record.  What is recorded instead: for four seeds, the moves of the first playout from the empty board up to the draw that
raises, and the position it raised in.  Data only.

    python tools/gen_simulate_golden.py     # needs the reference checkout (BOKEGO_REFERENCE, default /root/reference)
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import OUT, build_nets, game_record, go, mcts, torch  # noqa: E402  (seeds `random` before importing the reference)


def clear():
    mcts.MCTS._val_cache.clear(); mcts.MCTS._dist_cache.clear(); mcts.MCTS._fts_cache.clear()


def first_playout(pi, seed):
    clear()
    torch.manual_seed(seed)
    tree = mcts.MCTS(mcts.Go_MCTS(), pi, None, no_sim=False)
    node, moves, raised = tree.root, [], None
    while not node._terminal:
        try:
            nxt = node.find_random_child()
        except RuntimeError as e:
            raised = str(e)
            break
        moves.append(int(nxt.last_move))
        node = nxt
    color = go.BLACK if node.turn % 2 == 0 else go.WHITE
    ok = [m for m in node.get_legal_moves() if go.possible_eye(node.board, m) != color]
    return {"seed": seed, "moves": moves, "raised": raised, "stopped_at": game_record(node), "acceptable_moves_left": len(ok),
            "score": float(node.score())}


def searches(pi, v):
    out = []
    for vn in (None, v):
        for seed in range(10):
            clear()
            torch.manual_seed(seed)
            tree = mcts.MCTS(mcts.Go_MCTS(), pi, vn, no_sim=False, expand_thresh=3)
            try:
                tree.rollout(40)
                err = None
            except RuntimeError as e:
                err = str(e)
            out.append({"value_net": vn is not None, "seed": seed, "rollouts_done": int(tree.N[tree.root]), "error": err})
    return out


def main():
    pi, v = build_nets()[:2]
    rec = {"playouts": [first_playout(pi, s) for s in (5, 6, 7, 8)], "searches_of_40_rollouts": searches(pi, v)}
    for p in rec["playouts"]:
        print("seed", p["seed"], len(p["moves"]), "moves, raised:", p["raised"], "acceptable left:", p["acceptable_moves_left"])
    done = [s["rollouts_done"] for s in rec["searches_of_40_rollouts"]]
    print("searches: rollouts done before the error:", done, "; completed:", sum(s["error"] is None for s in rec["searches_of_40_rollouts"]))
    with open(os.path.join(OUT, "simulate_playouts.json"), "w") as f:
        json.dump(rec, f)


if __name__ == "__main__":
    main()
