#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Runs only where the reference checkout exists (env BOKEGO_REFERENCE, default
/root/reference); nothing here travels to the GPU box except its outputs, which
are pure data: positions, feature planes, network outputs, search traces and the
weight tensors re-serialised as BKW1.

    python tools/gen_golden.py            # everything
    python tools/gen_golden.py --skip-mcts

Seeds: random.seed(0) is set BEFORE importing bokego.go (its Zobrist table is
drawn at import, reference go.py:48-49, and set iteration order / tie breaks
depend on it).
"""
import argparse
import json
import os
import random
import sys
import time

REF = os.environ.get("BOKEGO_REFERENCE", "/root/reference")
if not os.path.isdir(os.path.join(REF, "bokego")):
    sys.exit(f"reference checkout not found at {REF}; set BOKEGO_REFERENCE")

random.seed(0)
sys.path.insert(0, REF)
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.set_grad_enabled(False)

import bokego.go as go  # noqa: E402
import bokego.mcts as mcts  # noqa: E402
import bokego.nnet as nnet  # noqa: E402

from bokego_amd.bkw import save_bkw, state_dict_to_tensors  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
W = os.path.join(REF, "data", "weights")
SGF_DIR = os.path.join(REF, "data", "bokevgnugo")


def build_nets():
    """policy = policy_19; value = ValueNet with policy_17 trunk + seeded head."""
    p19 = torch.load(os.path.join(W, "policy_19.pt"), map_location="cpu")["model_state_dict"]
    p17 = torch.load(os.path.join(W, "policy_17.pt"), map_location="cpu")["model_state_dict"]
    pi = nnet.PolicyNet()
    pi.load_state_dict(p19)
    pi.eval()
    torch.manual_seed(20260)
    v = nnet.ValueNet()
    v.load_policy_dict(p17)
    g = torch.Generator().manual_seed(20261)
    v.bn.running_mean.copy_(torch.randn(1, generator=g) * 0.5)
    v.bn.running_var.copy_(torch.rand(1, generator=g) * 1.5 + 0.5)
    v.bn.weight.copy_(torch.rand(1, generator=g) + 0.5)
    v.bn.bias.copy_(torch.randn(1, generator=g) * 0.3 + 0.5)
    v.lin_bn.running_mean.copy_(torch.randn(64, generator=g) * 0.5)
    v.lin_bn.running_var.copy_(torch.rand(64, generator=g) * 1.5 + 0.5)
    v.lin_bn.weight.copy_(torch.rand(64, generator=g) + 0.5)
    v.lin_bn.bias.copy_(torch.randn(64, generator=g) * 0.3)
    # widen the head so values spread over (-1,1) instead of hugging 0
    v.lin1.weight.mul_(2.0)
    v.lin2.weight.mul_(2.0)
    v.lin2.bias.sub_(0.8)
    v.eval()
    return pi, v


def game_record(g):
    return {"board": g.board, "ko": g.ko, "turn": g.turn, "last_move": g.last_move}


def feats_i8(g):
    f = nnet.features(g).numpy()
    assert f.shape == (27, 9, 9) and np.all(f == np.round(f)) and f.min() >= 0 and f.max() <= 7
    return f.astype(np.int8)


def fresh_copy(g):
    return go.Game(board=g.board, ko=g.ko, last_move=g.last_move, turn=g.turn)





def handmade_sequences():
    s = lambda r, c: 9 * r + c  # noqa: E731
    seqs = {}
    seqs["ko_centre"] = [s(3, 4), s(3, 5), s(5, 4), s(5, 5), s(4, 3), s(4, 6), s(0, 0), s(4, 4),
                         s(4, 5), s(8, 8), s(8, 0), s(4, 4), s(0, 8), s(7, 7), s(4, 5)]
    # corner capture, edge capture of a 2-stone chain, then a pass pair
    seqs["edge_caps"] = [s(0, 1), s(0, 0), s(1, 0), s(8, 8), s(0, 0), s(0, 4), s(1, 4), s(0, 5), s(1, 5),
                         s(7, 7), s(0, 3), s(6, 6), s(0, 6), go.PASS, go.PASS]
    # 3-stone chain touching the capturing move at two points (reference get_caps counts it twice)
    seqs["multi_caps"] = [s(2, 4), s(3, 4), s(3, 5), s(3, 3), s(2, 3), s(4, 3), s(3, 2), s(8, 8), s(4, 2),
                          s(8, 7), s(5, 3), s(8, 6), s(4, 4), s(3, 3), s(8, 0), s(3, 4)]
    # suicide / single-point-eye legality: black eye at corner, white may not play in
    seqs["eyes"] = [s(0, 1), s(5, 5), s(1, 0), s(5, 6), s(1, 1), s(5, 7), s(7, 0), s(4, 6), s(8, 1),
                    s(3, 3), s(7, 1), s(2, 2)]
    return seqs


def play_sequence(moves):
    g = go.Game(moves=[])
    recs = [(game_record(g), feats_i8(g), feats_i8(fresh_copy(g)))]
    for m in moves:
        g.play_move(m)
        recs.append((game_record(g), feats_i8(g), feats_i8(fresh_copy(g))))
    return recs


def eye_like(board, sq, color):
    """all on-board 4-neighbours are `color` stones (the build's own, bug-free, eye test)."""
    return all(board[n] == color for n in go.NEIGHBORS[sq])


def random_playout(seed, max_len=60):
    """BASELINE config-2 recipe: rng=default_rng(seed); L=rng.integers(0,61);
    play L uniformly random legal moves that do not fill an own single-point eye; pass if none."""
    rng = np.random.default_rng(seed)
    L = int(rng.integers(0, max_len + 1))
    g = go.Game(moves=[])
    nnet.features(g)  # touch liberties the way MCTS does at the root
    moves = []
    for _ in range(L):
        color = go.BLACK if g.turn % 2 == 0 else go.WHITE
        cands = [m for m in sorted(g.get_legal_moves()) if not eye_like(g.board, m, color)]
        m = cands[int(rng.integers(0, len(cands)))] if cands else go.PASS
        g.play_move(m)
        moves.append(m)
    return g, moves


def net_outputs(pi, v, feats, batch):
    x = torch.from_numpy(feats.astype(np.float32))
    logits, values = [], []
    for i in range(0, len(x), batch):
        xb = x[i:i + batch]
        logits.append(pi(xb).numpy())
        values.append(v(xb).numpy().reshape(-1))
    return np.concatenate(logits), np.concatenate(values)


def layer_acts(net, x):
    """post-ReLU activations of the 7 conv blocks, [7,128,9,9]"""
    acts = []
    h = x
    for i, m in enumerate(net.conv):
        h = m(h)
        if isinstance(m, torch.nn.ReLU):
            acts.append(h[0].numpy().copy())
    return np.stack(acts), h.reshape(-1).numpy().copy()


GTP_SESSION = [
    "protocol_version", "name", "version", "1 known_command genmove", "known_command frobnicate",
    "list_commands", "boardsize 9", "boardsize 19", "komi 6.5", "komi abc", "komi", "clear_board",
    "play b e5", "play w e5", "play b zz", "play", "genmove w", "showboard", "last_move", "move_history",
    "undo", "undo", "final_score", "play w c3", "play w g7", "last_move", "move_history", "final_score",
    "genmove b", "7 genmove b", "reg_genmove w", "move_history", "set_fixed_handicap 2", "clear_board",
    "set_fixed_handicap 9", "set_fixed_handicap 3", "showboard", "final_score", "genmove w", "last_move",
    "frobnicate now", "pondering maybe", "pondering off", "quit",
]


def gtp_transcript(pi, v):
    """Drive the reference's GTP.send() with a scripted session (200 rollouts per genmove)."""
    import bokego.gtp as rgtp
    mcts.MCTS._val_cache.clear(); mcts.MCTS._dist_cache.clear(); mcts.MCTS._fts_cache.clear()
    torch.manual_seed(0)
    g = rgtp.GTP(mcts.Go_MCTS(), pi, v, no_sim=True, time_lim=None, n_rollouts=200, pondering=False)
    g.running = True
    out = []
    for cmd in GTP_SESSION:
        out.append([cmd, g.send(cmd)])
    return out


def whole_game_trace(pi, v, n_roll=1600, max_moves=100):
    """One WHOLE game of the reference's search against itself: n_roll rollouts before every move, choose() until the root
    is terminal (turn > MAX_TURNS, mcts.py:362-364) or has no child -- the regime in which ms/move over 80-move games is
    quoted.  The tree and its statistics are re-used across moves as GTP.genmove does (mcts.py:110-131)."""
    mcts.MCTS._val_cache.clear(); mcts.MCTS._dist_cache.clear(); mcts.MCTS._fts_cache.clear()
    torch.manual_seed(0)
    tree = mcts.MCTS(mcts.Go_MCTS(), pi, v, no_sim=True)
    moves = []
    t0 = time.time()
    while len(moves) < max_moves:
        root = tree.root
        if root._terminal or not tree.children.get(root):
            break
        tree.rollout(n_roll)
        kids = {int(c.last_move): int(tree.N[c]) for c in tree.children[root]}
        rootN = int(tree.N[root]); wr = float(tree.winrate())
        best = tree.choose()
        moves.append({"move": int(best.last_move), "alpha": go.unsquash(best.last_move), "root_N": rootN,
                      "root_winrate": wr, "child_N": kids, "best_V": float(tree.V[best])})
        print(len(moves), moves[-1]["alpha"], rootN, max(kids.values()), f"{time.time() - t0:.0f}s", flush=True)
    out = {"r%d_game" % n_roll: {"rollouts": n_roll, "kwargs": {}, "moves": moves,
                                  "final_board": tree.root.board, "final_turn": int(tree.root.turn),
                                  "n_value_evals": len(mcts.MCTS._val_cache),
                                  "n_policy_evals": len(mcts.MCTS._dist_cache)}}
    with open(os.path.join(OUT, "mcts_trace_game.json"), "w") as f:
        json.dump(out, f)
    print("whole game:", len(moves), "moves in", f"{time.time() - t0:.0f}s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only-gtp", action="store_true")
    ap.add_argument("--skip-mcts", action="store_true")
    ap.add_argument("--only-game", action="store_true",
                    help="only the whole-game 1600-rollout trace (tests/golden/mcts_trace_game.json)")
    ap.add_argument("--game-rollouts", type=int, default=1600)
    ap.add_argument("--playouts", type=int, default=256)
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    pi, v = build_nets()
    if args.only_game:
        whole_game_trace(pi, v, args.game_rollouts)
        return
    if args.only_gtp:
        with open(os.path.join(OUT, "gtp_transcript.json"), "w") as f:
            json.dump({"n_rollouts": 200, "session": gtp_transcript(pi, v)}, f, indent=0)
        return

    # ---- weights --------------------------------------------------------------------
    save_bkw(os.path.join(OUT, "policy_19.bkw"), state_dict_to_tensors(pi.state_dict()))
    save_bkw(os.path.join(OUT, "value_synth.bkw"), state_dict_to_tensors(v.state_dict()))

    # ---- G1/G2: SGF positions + features --------------------------------------------
    recs, finc, ffresh, game_of = [], [], [], []
    sgf_moves = {}
    for gi in range(1, 11):
        path = os.path.join(SGF_DIR, f"boke_gnugo_{gi}.sgf")
        g = go.Game(sgf=path)
        sgf_moves[f"boke_gnugo_{gi}"] = list(g.moves)
        if gi == 1:
            recs.append(game_record(g)); finc.append(feats_i8(g)); ffresh.append(feats_i8(fresh_copy(g)))
            game_of.append(0)
        for _ in range(len(g.moves)):
            g.play_move()
            recs.append(game_record(g)); finc.append(feats_i8(g)); ffresh.append(feats_i8(fresh_copy(g)))
            game_of.append(gi)
    n_sgf = len(recs)
    hand = handmade_sequences()
    hand_index = {}
    for name, mv in hand.items():
        r = play_sequence(mv)
        hand_index[name] = [len(recs), len(recs) + len(r)]
        for rec, fi, ff in r:
            recs.append(rec); finc.append(fi); ffresh.append(ff); game_of.append(-1)
    finc = np.stack(finc); ffresh = np.stack(ffresh)
    differs = np.any(finc.reshape(len(finc), -1) != ffresh.reshape(len(finc), -1), axis=1)
    print(f"positions: {len(recs)} ({n_sgf} from SGF); stale-liberty planes differ on {int(differs.sum())}")
    with open(os.path.join(OUT, "positions.json"), "w") as f:
        json.dump({"n_sgf": n_sgf, "positions": recs, "game_of": game_of, "sgf_moves": sgf_moves,
                   "handmade_moves": hand, "handmade_index": hand_index}, f)
    np.savez_compressed(os.path.join(OUT, "features.npz"), incremental=finc, fresh=ffresh)

    # ---- G3: net outputs -------------------------------------------------------------
    torch.manual_seed(0)
    lg1, va1 = net_outputs(pi, v, finc, 1)
    lg64, va64 = net_outputs(pi, v, finc, 64)
    probs = torch.softmax(torch.from_numpy(lg1), dim=1)
    cat = torch.distributions.Categorical(probs).probs.numpy()
    print("batch1-vs-batch64 max|dlogit| %.3g  max|dvalue| %.3g" %
          (np.abs(lg1 - lg64).max(), np.abs(va1 - va64).max()))
    print("value range", va1.min(), va1.max(), "logit range", lg1.min(), lg1.max())
    np.savez_compressed(os.path.join(OUT, "nets.npz"), logits_b1=lg1, values_b1=va1, logits_b64=lg64,
                        values_b64=va64, probs_b1=probs.numpy(), categorical_probs_b1=cat)
    idx = [0, 200]
    pol_acts, val_acts, pol_head, val_head = [], [], [], []
    for i in idx:
        x = torch.from_numpy(finc[i:i + 1].astype(np.float32))
        a, h = layer_acts(pi, x); pol_acts.append(a); pol_head.append(h)
        a, h = layer_acts(v, x); val_acts.append(a); val_head.append(h)
    np.savez_compressed(os.path.join(OUT, "layers.npz"), index=np.array(idx), policy=np.stack(pol_acts),
                        value=np.stack(val_acts), policy_head=np.stack(pol_head), value_head=np.stack(val_head))

    # ---- random playouts (config-2 input recipe), pins the build's own board engine ----
    pm, pf, pr = [], [], []
    t0 = time.time()
    for i in range(args.playouts):
        g, moves = random_playout(20260 + i)
        pm.append(moves); pf.append(feats_i8(g)); pr.append(game_record(g))
    pf = np.stack(pf)
    plg, pva = net_outputs(pi, v, pf, 64)
    print(f"playouts: {len(pm)} in {time.time() - t0:.1f}s")
    with open(os.path.join(OUT, "playouts.json"), "w") as f:
        json.dump({"seed_base": 20260, "moves": pm, "final": pr}, f)
    np.savez_compressed(os.path.join(OUT, "playouts.npz"), features=pf, logits=plg, values=pva)

    # ---- G4: MCTS traces ---------------------------------------------------------------
    if not args.skip_mcts:
        traces = {}
        for name, n_roll, n_moves, kw in [("r1600", 1600, 10, {}), ("r300_t20", 300, 6, {"expand_thresh": 20})]:
            mcts.MCTS._val_cache.clear(); mcts.MCTS._dist_cache.clear(); mcts.MCTS._fts_cache.clear()
            torch.manual_seed(0)
            tree = mcts.MCTS(mcts.Go_MCTS(), pi, v, no_sim=True, **kw)
            moves = []
            t0 = time.time()
            for _ in range(n_moves):
                tree.rollout(n_roll)
                root = tree.root
                kids = {int(c.last_move): int(tree.N[c]) for c in tree.children[root]}
                vsum = {int(c.last_move): float(tree.V[c]) for c in tree.children[root]}
                rootN = int(tree.N[root]); wr = float(tree.winrate())
                best = tree.choose()
                moves.append({"move": int(best.last_move), "alpha": go.unsquash(best.last_move), "root_N": rootN,
                              "root_winrate": wr, "child_N": kids, "child_V": vsum})
            traces[name] = {"rollouts": n_roll, "kwargs": kw, "moves": moves,
                            "n_value_evals": len(mcts.MCTS._val_cache), "n_policy_evals": len(mcts.MCTS._dist_cache)}
            print(name, [m["alpha"] for m in moves], [max(m["child_N"].values()) for m in moves],
                  f"{time.time() - t0:.1f}s", traces[name]["n_value_evals"], traces[name]["n_policy_evals"])
        with open(os.path.join(OUT, "mcts_trace.json"), "w") as f:
            json.dump(traces, f)
    with open(os.path.join(OUT, "gtp_transcript.json"), "w") as f:
        json.dump({"n_rollouts": 200, "session": gtp_transcript(pi, v)}, f, indent=0)


if __name__ == "__main__":
    main()
