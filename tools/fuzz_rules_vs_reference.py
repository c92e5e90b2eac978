#!/usr/bin/env python3
"""Random games on the REFERENCE's rules engine (bokego/go.py, nnet.features) and on the native board (libbkgo.so) side by side:
after every move -- legal and illegal attempts, passes -- the board, ko, turn, legal-move set, liberty cache, score, one-point-eye
colours and all 27 feature planes (incremental mode: the history-dependent liberty cache included) must be equal.  A check to run
where the reference checkout is (BOKEGO_REFERENCE); nothing of it travels.
    python tools/fuzz_rules_vs_reference.py [seed] [seconds]"""
import os
import random
import sys
import time

REF = os.environ.get("BOKEGO_REFERENCE", "/root/reference")
if not os.path.isdir(os.path.join(REF, "bokego")):
    sys.exit(f"reference checkout not found at {REF}; set BOKEGO_REFERENCE")
random.seed(0)
sys.path.insert(0, REF)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bokego.go as rgo  # noqa: E402
import bokego.nnet as rnnet  # noqa: E402

from bokego_amd import go  # noqa: E402

rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 60)
lib = go.golib()
games = moves = illegal = 0
while time.time() < t_end:
    r, g = rgo.Game(moves=[]), go.Game(moves=[])
    check_feats_every = rng.choice([1, 2, 5])
    for ply in range(rng.randint(5, 90)):
        mv = go.PASS if rng.random() < 0.03 else rng.randrange(81)
        if rng.random() < 0.7:                       # mostly legal moves, so that games get long
            legal = sorted(r.get_legal_moves())
            if legal:
                mv = rng.choice(legal)
        try:
            r.play_move(mv) if mv != go.PASS else r.play_pass()
            r_ok = True
        except rgo.IllegalMove:
            r_ok = False
        try:
            g.play_move(mv) if mv != go.PASS else g.play_pass()
            g_ok = True
        except go.IllegalMove:
            g_ok = False
        assert r_ok == g_ok, ("legality", games, ply, mv, r_ok, g_ok, r.board)
        moves += 1
        illegal += not r_ok
        assert g.board == r.board and g.ko == r.ko and g.turn == r.turn and g.last_move == r.last_move, ("state", games, ply, mv)
        assert sorted(g.get_legal_moves()) == sorted(r.get_legal_moves()), ("legal set", games, ply)
        assert g.score() == r.score(), ("score", games, ply)
        if ply % check_feats_every == 0:
            fr = rnnet.features(r).numpy().astype(np.uint8)           # refreshes the reference's liberty cache as MCTS does
            fg = g.features_u8()
            assert np.array_equal(fr, fg), ("features", games, ply, np.argwhere(fr != fg)[:5])
            eye = {None: 0, rgo.BLACK: 1, rgo.WHITE: 2}
            assert [eye[rgo.possible_eye(r.board, s)] for s in range(81)] == [lib.bk_pos_possible_eye(ctypes.byref(g._pos), s) for s in range(81)]
        if r.turn > 100:
            break
    games += 1
print(f"{games} games, {moves} moves ({illegal} illegal attempts): equal")
