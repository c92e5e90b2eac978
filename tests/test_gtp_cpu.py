"""The GTP front-end replayed against a transcript recorded from the reference's GTP.send()
(tests/golden/gtp_transcript.json; 200 rollouts per genmove), on CPU with the oracle nets."""
import io
import json
import os

import pytest
import torch

from bokego_amd.bkw import load_bkw
from bokego_amd.gtp import GTP
from bokego_amd.mcts import Go_MCTS
from oracle.oracle import OraclePolicy, OracleValue

from conftest import GOLDEN


class _Wrap:
    def __init__(self, fn, value=False):
        self.fn, self.value = fn, value

    def to(self, d):
        return self

    def __call__(self, x):
        o = self.fn(x.numpy())
        return torch.from_numpy(o.reshape(-1, 1) if self.value else o)


@pytest.fixture(scope="module")
def nets():
    return (_Wrap(OraclePolicy(load_bkw(os.path.join(GOLDEN, "policy_19.bkw")))),
            _Wrap(OracleValue(load_bkw(os.path.join(GOLDEN, "value_synth.bkw"))), True))


def test_replies_match_reference_transcript(nets):
    t = json.load(open(os.path.join(GOLDEN, "gtp_transcript.json")))
    torch.manual_seed(0)
    g = GTP(Go_MCTS(), nets[0], nets[1], no_sim=True, time_lim=None, n_rollouts=t["n_rollouts"])
    g.running = True
    for cmd, want in t["session"]:
        got = g.send(cmd)
        assert got == want, (cmd, got, want)
    assert g.running is False            # quit


def test_stdin_loop_and_clear_cache(nets):
    g = GTP(Go_MCTS(), nets[0], nets[1], no_sim=True, time_lim=None, n_rollouts=20, expand_thresh=5)
    out = io.StringIO()
    g.start(io.StringIO("# comment\nname\n\n3 play b e5\nclear_cache\ngenmove w\nfinal_score\nquit\nname\n"), out)
    lines = out.getvalue().split("\n\n")
    assert lines[0] == "= boke" and lines[1] == "=3 " and lines[2] == "= " and lines[3].startswith("= ")
    assert lines[4].startswith("= ") and lines[5] == "= " and len(lines) == 7   # nothing after quit
    assert len(g.genmove_seconds) == 1
