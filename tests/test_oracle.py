"""Pins the CPU oracle (oracle/nnet_ref.c) against vectors produced by the reference itself."""
import os
import time

import numpy as np
import pytest

from bokego_amd.bkw import load_bkw
from oracle.oracle import OraclePolicy, OracleValue

from conftest import GOLDEN

TOL_LOGIT = 1e-4   # north_star tolerance (fp32, logits reach |50|)
TOL_VALUE = 1e-4
TOL_PROB = 1e-5


@pytest.fixture(scope="module")
def nets():
    return (OraclePolicy(load_bkw(os.path.join(GOLDEN, "policy_19.bkw"))),
            OracleValue(load_bkw(os.path.join(GOLDEN, "value_synth.bkw"))))


@pytest.fixture(scope="module")
def gold():
    f = np.load(os.path.join(GOLDEN, "features.npz"))["incremental"].astype(np.float32)
    n = np.load(os.path.join(GOLDEN, "nets.npz"))
    return f, n


def test_known_answers_empty_board(nets, gold):
    # SURVEY 8c known answers for policy_19 on the empty board
    f, n = gold
    assert f[0].sum() == 531
    lg, pr = nets[0](f[:1], want_probs=True)
    assert abs(lg[0, 40] - 12.709958) < 1e-4 and abs(lg.max() - 12.709958) < 1e-4
    assert abs(lg.min() + 15.542190) < 1e-4
    assert abs(pr[0, 40] - 0.818645) < 1e-5


def test_policy_matches_reference(nets, gold):
    f, n = gold
    lg, pr = nets[0](f, want_probs=True)
    assert np.abs(lg - n["logits_b1"]).max() < TOL_LOGIT
    assert np.abs(pr - n["probs_b1"]).max() < TOL_PROB
    assert np.array_equal(lg.argmax(1), n["logits_b1"].argmax(1))


def test_value_matches_reference(nets, gold):
    f, n = gold
    v = nets[1](f)
    assert np.abs(v - n["values_b1"]).max() < TOL_VALUE


def test_layer_activations(nets):
    L = np.load(os.path.join(GOLDEN, "layers.npz"))
    f = np.load(os.path.join(GOLDEN, "features.npz"))["incremental"].astype(np.float32)[L["index"]]
    _, acts = nets[0](f, want_acts=True)
    assert np.abs(acts - L["policy"]).max() < 1e-4
    _, vacts = nets[1](f, want_acts=True)
    assert np.abs(vacts - L["value"]).max() < 1e-4
    # the two trunks are different networks (catches accidental trunk sharing)
    assert np.abs(L["policy"] - L["value"]).max() > 1e-2


def test_playout_positions(nets):
    p = np.load(os.path.join(GOLDEN, "playouts.npz"))
    f = p["features"].astype(np.float32)
    assert np.abs(nets[0](f) - p["logits"]).max() < TOL_LOGIT
    assert np.abs(nets[1](f) - p["values"]).max() < TOL_VALUE


def test_hard_positions(nets):
    """Reference outputs on the 48 hardest of 49,152 random-playout positions (|logit| up to 65)."""
    h = np.load(os.path.join(GOLDEN, "hard_positions.npz"))
    f = h["features"].astype(np.float32)
    assert np.abs(nets[0](f) - h["logits"]).max() < TOL_LOGIT
    assert np.abs(nets[1](f) - h["values"]).max() < TOL_VALUE


def test_torch_restatement_matches_reference(gold):
    """oracle/torch_ref.py (the reference's ops re-assembled on torch CPU; bench.py's reference-CPU baseline) against
    the reference's own outputs: B=1 and batched, policy logits / softmax / value."""
    import torch
    from oracle.torch_ref import TorchPolicy, TorchValue, leaf_eval
    f, n = gold
    P = TorchPolicy(load_bkw(os.path.join(GOLDEN, "policy_19.bkw")))
    V = TorchValue(load_bkw(os.path.join(GOLDEN, "value_synth.bkw")))
    x = torch.from_numpy(f)
    lg, pr, va = leaf_eval(P, V, x)
    assert np.abs(lg.numpy() - n["logits_b64"]).max() < TOL_LOGIT and np.abs(lg.numpy() - n["logits_b1"]).max() < TOL_LOGIT
    assert np.abs(pr.numpy() - n["probs_b1"]).max() < TOL_PROB
    assert np.abs(va.numpy() - n["values_b1"]).max() < TOL_VALUE
    for i in (0, 17, 300, 535):
        l1, _, v1 = leaf_eval(P, V, x[i:i + 1])
        assert np.abs(l1.numpy() - n["logits_b1"][i]).max() < TOL_LOGIT and abs(float(v1) - n["values_b1"][i]) < TOL_VALUE
    h = np.load(os.path.join(GOLDEN, "hard_positions.npz"))
    lg, _, va = leaf_eval(P, V, torch.from_numpy(h["features"].astype(np.float32)))
    assert np.abs(lg.numpy() - h["logits"]).max() < TOL_LOGIT and np.abs(va.numpy() - h["values"]).max() < TOL_VALUE


@pytest.mark.parametrize("wset", ["A", "B"])
def test_sweep_worst_positions_both_weight_sets(wset):
    """The oracle (C and torch restatements) on the sweep's worst positions, both weight sets: the second set pins the
    restatements on weights they were not developed against."""
    import torch
    from oracle.torch_ref import TorchPolicy, TorchValue
    w = np.load(os.path.join(GOLDEN, "sweep_worst.npz"))
    p19, vs = load_bkw(os.path.join(GOLDEN, "policy_19.bkw")), load_bkw(os.path.join(GOLDEN, "value_synth.bkw"))
    if wset == "A":
        pw, vw = p19, vs
    else:
        head_b = np.load(os.path.join(GOLDEN, "value_head_b.npz"))
        pw, vw = {k: v for k, v in vs.items() if k.startswith("conv.")}, dict(p19)
        vw.update({k: head_b[k] for k in head_b.files})
    f = w[f"features_{wset}"].astype(np.float32)
    assert np.abs(OraclePolicy(pw)(f) - w[f"logits_{wset}"]).max() < TOL_LOGIT
    assert np.abs(OracleValue(vw)(f) - w[f"values_{wset}"]).max() < TOL_VALUE
    assert np.abs(TorchPolicy(pw)(torch.from_numpy(f)).numpy() - w[f"logits_{wset}"]).max() < TOL_LOGIT
    assert np.abs(TorchValue(vw)(torch.from_numpy(f)).numpy() - w[f"values_{wset}"]).max() < TOL_VALUE
