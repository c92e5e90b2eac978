"""NativeMCTS / NativeGTP (native tree under the reference's single-tree surface) against the Python
driver and against the transcript recorded from the reference."""
import json
import os

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
