"""The native board + feature encoder (libbkgo.so) against positions, feature planes and random
playouts recorded from the reference's go.Game / nnet.features() (tests/golden, tools/gen_golden.py)."""
import json
import os

import numpy as np
import pytest

from bokego_amd import go

from conftest import GOLDEN


@pytest.fixture(scope="module")
def gold():
    pos = json.load(open(os.path.join(GOLDEN, "positions.json")))
    f = np.load(os.path.join(GOLDEN, "features.npz"))
    return pos, f["incremental"].astype(np.uint8), f["fresh"].astype(np.uint8)


def _same_state(g, rec):
    assert g.board == rec["board"] and g.ko == rec["ko"] and g.turn == rec["turn"] and g.last_move == rec["last_move"]


def _replay(moves, recs, finc, ffresh, start):
    g = go.Game(moves=[])
    i = start
    if recs[i]["turn"] == 0:
        _same_state(g, recs[i])
        assert np.array_equal(g.features_u8(), finc[i])
        i += 1
    for m in moves:
        g.play_move(m)
        _same_state(g, recs[i])
        assert np.array_equal(g.features_u8(), finc[i]), f"incremental planes differ at position {i}"
        assert np.array_equal(g.features_u8(fresh=True), ffresh[i]), f"fresh planes differ at position {i}"
        h = go.Game(board=recs[i]["board"], ko=recs[i]["ko"], last_move=recs[i]["last_move"], turn=recs[i]["turn"])
        assert np.array_equal(h.features_u8(), ffresh[i])
        assert hash(h) == hash(g)
        i += 1
    return i


def test_sgf_games_positions_and_planes(gold):
    pos, finc, ffresh = gold
    recs = pos["positions"]
    i = 0
    for gi in range(1, 11):
        i = _replay(pos["sgf_moves"][f"boke_gnugo_{gi}"], recs, finc, ffresh, i)
    assert i == pos["n_sgf"] == 474
    # the stale-liberty quirk is really exercised: many positions differ between the two modes
    assert sum(not np.array_equal(a, b) for a, b in zip(finc[:474], ffresh[:474])) > 100


def test_handmade_ko_capture_eye_positions(gold):
    pos, finc, ffresh = gold
    for name, (lo, hi) in pos["handmade_index"].items():
        end = _replay(pos["handmade_moves"][name], pos["positions"], finc, ffresh, lo)
        assert end == hi, name


def test_known_answers():
    g = go.Game()
    assert g.features_u8().sum() == 531
    assert len(g.get_legal_moves()) == 81 and g.score() == -5.5
    # ko: retaking immediately is illegal with rule_type "ko"
    for m in [31, 32, 49, 50, 39, 42, 0, 40, 41]:
        g.play_move(m)
    assert g.ko == 40 and not g.is_legal(40)
    with pytest.raises(go.IllegalMove) as e:
        g.play_move(40)
    assert e.value.rule_type == "ko"
    with pytest.raises(go.IllegalMove) as e:
        g.play_move(41)
    assert e.value.rule_type == "not_empty"
    before = g.key()
    with pytest.raises(go.IllegalMove):
        g.play_move(41)
    assert g.key() == before                      # failed moves leave the state untouched
    # suicide in a corner eye
    s = go.Game()
    for m in [1, 40, 9, 41]:
        s.play_move(m)
    s.play_move(go.PASS)
    with pytest.raises(go.IllegalMove) as e:
        s.play_move(0)
    assert e.value.rule_type == "suicide"
    assert go.squash("E5") == 40 and go.unsquash(40) == "E5" and go.squash("J9") == 80 and go.squash("pass") == go.PASS


def test_caps_plane_double_counts_like_reference(gold):
    pos, finc, _ = gold
    lo, _ = pos["handmade_index"]["multi_caps"]
    i = lo + 12                                   # black to play E5 capturing a 3-stone chain touching at 2 points
    assert pos["positions"][i]["turn"] == 12
    assert finc[i][20 + 5, 4, 4] == 6             # 3 stones counted twice -> plane "6 captures"


def test_random_playout_recipe_matches_reference():
    from bokego_amd.workload import random_playout
    pl = json.load(open(os.path.join(GOLDEN, "playouts.json")))
    feats = np.load(os.path.join(GOLDEN, "playouts.npz"))["features"].astype(np.uint8)
    for i, mv in enumerate(pl["moves"]):
        g, moves = random_playout(pl["seed_base"] + i)
        assert moves == mv, f"playout {i}"
        _same_state(g, pl["final"][i])
        assert np.array_equal(g.features_u8(), feats[i])


def test_copy_is_independent():
    g = go.Game(moves=[])
    g.play_move(40)
    h = g.copy()
    h.play_move(41)
    assert g.turn == 1 and h.turn == 2 and g.moves == [40] and h.moves == [40, 41] and g != h
    import copy
    assert copy.deepcopy(g) == g


def test_score_area():
    g = go.Game()
    for m in [36, 44, 37, 43, 38, 42, 39, 41]:    # incomplete walls: regions touch both colours -> neutral
        g.play_move(m)
    assert g.area_score() == 4 - (4 + 5.5)
    assert g.score() == 0 - (0 + 5.5)             # reference quirk: stones bordering neutral points vanish
    b = "X" * 36 + "." * 9 + "O" * 36
    assert go.Game(board=b).score() == 36 - (36 + 5.5)
    b = "X" * 36 + "." * 45
    assert go.Game(board=b).score() == 81 - 5.5


def test_children_without_a_flood_fill_per_move_equal_the_definition():
    """bk_pos_children finds a position's chains once and builds the successors of non-capturing moves directly; the definition
    is one bk_pos_play per empty point (bk_pos_children_slow).  Record for record -- board, ko, hash, turn, last move, liberty
    cache -- over random playouts (captures, kos, suicides, passes, stale liberty caches included), 20,000+ positions."""
    import ctypes
    lib = go.golib()
    rng = np.random.default_rng(5)
    checked = caps = 0
    for game in range(400):
        pos = go.Pos()
        lib.bk_pos_init(ctypes.byref(pos))
        for ply in range(int(rng.integers(5, 110))):
            a, b = (go.Pos * 81)(), (go.Pos * 81)()
            ma, mb = (ctypes.c_int16 * 81)(), (ctypes.c_int16 * 81)()
            na = lib.bk_pos_children(ctypes.byref(pos), a, ma)
            nb = lib.bk_pos_children_slow(ctypes.byref(pos), b, mb)
            assert na == nb and list(ma[:na]) == list(mb[:nb])
            assert bytes(a)[:192 * na] == bytes(b)[:192 * nb]
            checked += 1
            if na == 0:
                break
            stones = sum(1 for c in bytes(pos)[:81] if c)
            k = int(rng.integers(0, na))
            if rng.random() < 0.03:
                lib.bk_pos_play(ctypes.byref(pos), go.PASS)
            else:
                pos = go.Pos.from_buffer_copy(a[k])
                caps += sum(1 for c in bytes(pos)[:81] if c) <= stones
            if rng.random() < 0.2:                      # a feature request refreshes the liberty cache in between
                libs = (ctypes.c_uint8 * 81)()
                lib.bk_pos_liberties(ctypes.byref(pos), libs)
    assert checked > 20000 and caps > 300


def test_possible_eye_matches_the_reference_table_bug_included():
    """go.possible_eye (go.py:470-485) on 1,392 boards x 81 points recorded from the reference (tools/gen_eye_golden.py): the
    one-point-eye test Go_MCTS.get_move uses (mcts.py:354), with the reference's DIAGONALS table (go.py:372-373), which
    looks at (x-1,y-1) twice and never at (x-1,y+1)."""
    import ctypes
    z = np.load(os.path.join(GOLDEN, "possible_eye.npz"))
    lib = go.golib()
    n_eyes = 0
    for b, want in zip(z["boards"], z["eyes"]):
        g = go.Game(board="".join(".XO"[c] for c in b))
        got = np.array([lib.bk_pos_possible_eye(ctypes.byref(g._pos), s) for s in range(81)], np.int8)
        assert (got == want).all(), ("".join(".XO"[c] for c in b), np.nonzero(got != want)[0])
        n_eyes += int((want > 0).sum())
    assert n_eyes > 500
    # the blind corner: black stones around E5 and white stones on the two diagonals the table does see twice / never
    board = ["."] * 81
    for s in (3 * 9 + 4, 5 * 9 + 4, 4 * 9 + 3, 4 * 9 + 5):
        board[s] = "X"
    board[3 * 9 + 5] = "O"          # (x-1, y+1): never looked at
    board[5 * 9 + 5] = "O"          # (x+1, y+1): one fault -> still an eye
    g = go.Game(board="".join(board))
    assert lib.bk_pos_possible_eye(ctypes.byref(g._pos), 40) == 1
    board[3 * 9 + 3] = "O"          # (x-1, y-1): counted twice -> three faults
    g = go.Game(board="".join(board))
    assert lib.bk_pos_possible_eye(ctypes.byref(g._pos), 40) == 0
