"""Synthetic position generator of BASELINE config 2: seeded random legal playouts."""
import numpy as np

from . import go


def random_playout(seed, max_len=60):
    """rng = default_rng(seed); L = rng.integers(0, max_len+1); from the empty board play L uniformly
    random legal moves that do not fill an own single-point eye (ascending candidate order; pass if
    none).  Returns (game, moves).  Pinned move-for-move against the reference's rules engine by
    tests/golden/playouts.json."""
    rng = np.random.default_rng(seed)
    L = int(rng.integers(0, max_len + 1))
    g = go.Game(moves=[])
    g.get_liberties()  # MCTS touches the root's liberties before the first move (mcts.py:153-157)
    lib = go.golib()
    import ctypes
    moves = []
    for _ in range(L):
        color = 1 if g.turn % 2 == 0 else 2
        cands = [m for m in g.get_legal_moves() if not lib.bk_pos_eye_like(ctypes.byref(g._pos), m, color)]
        m = cands[int(rng.integers(0, len(cands)))] if cands else go.PASS
        g.play_move(m)
        moves.append(m)
    return g, moves


def make_batch(B, seed_base=20260, dtype=np.float32, with_records=False):
    """[B,27,9,9] feature planes of B random-playout positions (seeds seed_base .. seed_base+B-1); with
    with_records also the same positions as uint8 [B,192] records (liberty cache refreshed), the input of
    LeafEngine.submit_positions."""
    out = np.empty((B, 27, 9, 9), np.uint8)
    recs = np.empty((B, 192), np.uint8)
    for i in range(B):
        g, _ = random_playout(seed_base + i)
        out[i] = g.features_u8()
        recs[i] = np.frombuffer(bytes(g._pos), np.uint8)   # features_u8 has refreshed the cache
    return (out.astype(dtype), recs) if with_records else out.astype(dtype)
