"""NativeMCTS / NativeGTP (native tree under the reference's single-tree surface) against the Python
driver and against the transcript recorded from the reference."""
import json
import os

import numpy as np
import pytest
import torch

from bokego_amd import go
from bokego_amd.bkw import load_bkw
from bokego_amd.gtp import GTP, NativeGTP
from bokego_amd.mcts import MCTS, Go_MCTS
from bokego_amd.mcts_native import NativeMCTS, Position

from conftest import GOLDEN
from test_selfplay_cpu import FakeNets, _Wrap


def test_native_equals_python_tree_including_outside_moves():
    f = FakeNets()
    py = MCTS(Go_MCTS(), _Wrap(f.policy), _Wrap(f.value, True), expand_thresh=6)
    nat = NativeMCTS(Position(), _Wrap(f.policy), _Wrap(f.value, True), expand_thresh=6)
    outside = {2: 10, 5: 70}                    # plies at which the "opponent" plays a fixed move instead
    for ply in range(8):
        py.rollout(150); nat.rollout(150)
        want = {c.mv: (py.N[c], py.V[c]) for c in py.children[py.root]}
        assert nat.child_stats() == want
        assert abs(nat.winrate() - py.winrate()) < 1e-12
        if ply in outside:
            mv = outside[ply]
            py.set_root(py.root.make_move(mv)); nat.set_root(nat.root.make_move(mv))
        else:
            a, b = py.choose(), nat.choose()
            assert a.last_move == b.last_move
        assert nat.root.key() == py.root.key()
    with pytest.raises(go.IllegalMove):
        nat.play(nat.root.last_move)            # occupied point
    nat.play(go.PASS)
    assert nat.root._terminal and nat.choose().key() == nat.root.key()
    with pytest.raises(TypeError):
        NativeMCTS(Position())


def test_native_gtp_matches_reference_transcript():
    from oracle.oracle import OraclePolicy, OracleValue
    P = _Wrap(OraclePolicy(load_bkw(os.path.join(GOLDEN, "policy_19.bkw"))))
    V = _Wrap(OracleValue(load_bkw(os.path.join(GOLDEN, "value_synth.bkw"))), True)
    t = json.load(open(os.path.join(GOLDEN, "gtp_transcript.json")))
    g = NativeGTP(Position(), P, V, no_sim=True, time_lim=None, n_rollouts=t["n_rollouts"])
    g.running = True
    for cmd, want in t["session"]:
        assert g.send(cmd) == want, cmd


@pytest.mark.parametrize("prune", [0, 1])
def test_speculative_evaluation_does_not_change_the_search(prune):
    """search_params.speculate: leaves that reach N visits get their policy and their would-be children's values
    evaluated with the next request that goes out anyway.  The search is the same search -- chosen moves, every root
    child's (N, V) after every move -- with fewer requests (and more evaluations)."""
    f = FakeNets()
    # row by row: a BLAS matmul's bits depend on the batch's shape, and speculation changes which rows share a batch
    # (the HIP engine's outputs do not depend on the batch: tests/test_gpu_parity.py)
    pol = lambda x: np.stack([f.policy(r[None])[0] for r in x]) if len(x) else np.zeros((0, 81), np.float32)  # noqa: E731
    val = lambda x: np.array([f.value(r[None])[0] for r in x], np.float32)  # noqa: E731
    trees = {s: NativeMCTS(Position(), _Wrap(pol), _Wrap(val, True), expand_thresh=20, speculate=s, speculate_rows=256, prune=prune)
             for s in (0, 8, 15)}
    for ply in range(10):
        stats = {}
        for s, t in trees.items():
            t.rollout(300)
            stats[s] = t.child_stats()
            t.choose()
        assert stats[8] == stats[0] and stats[15] == stats[0], ply
    keys = {s: t.root.key() for s, t in trees.items()}
    assert keys[8] == keys[0] and keys[15] == keys[0]
    info = {s: t._pool.info(0) for s, t in trees.items()}
    assert info[8]["n_requests"] < info[0]["n_requests"] and info[15]["n_requests"] < info[0]["n_requests"]
    assert info[8]["n_value_evals"] >= info[0]["n_value_evals"]
