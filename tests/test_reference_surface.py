"""The reference side of the drop-in claim, asserted from a recorded file.

tools/check_reference_surface.py ran -- in the build container, where the reference can be imported -- the REFERENCE's own
`MCTS` (bokego/mcts.py:46-79) and `GTP` (bokego/gtp.py:47-55) classes on this repo's shim networks, built the way
boke.py:30-38 builds its nets, and replayed the reference's recorded traces and GTP session.  The reference cannot travel, so
the outcome is committed as data (tests/golden/reference_surface.json) and checked here; the hash ties the file to the
shim source it was made with (change bokego_amd/nnet.py -> re-run the tool)."""
import hashlib
import json
import os

from conftest import GOLDEN, REPO


def test_reference_mcts_and_gtp_ran_unchanged_on_the_shims():
    d = json.load(open(os.path.join(GOLDEN, "reference_surface.json")))
    assert d["reference_classes"] == ["bokego.mcts.MCTS", "bokego.gtp.GTP"]
    assert d["shim_classes"] == ["bokego_amd.nnet.HipPolicyNet", "bokego_amd.nnet.HipValueNet"]
    sha = hashlib.sha256(open(os.path.join(REPO, "bokego_amd", "nnet.py"), "rb").read()).hexdigest()
    assert d["shim_sha256"] == sha, "bokego_amd/nnet.py changed: re-run tools/check_reference_surface.py"
    gold = json.load(open(os.path.join(GOLDEN, "mcts_trace.json")))
    assert set(d["traces"]) == {"r300_t20", "r1600"}
    for name, t in d["traces"].items():
        assert t["all_equal"] and t["rollouts"] == gold[name]["rollouts"] and len(t["moves"]) == len(gold[name]["moves"])
        for got, want in zip(t["moves"], gold[name]["moves"]):
            assert got["move"] == want["move"] and got["move_equal"] and got["child_N_equal"] and got["root_winrate_delta"] < 1e-4
        # the reference made exactly as many network calls on the shims as on its own nets
        assert t["n_value_evals"] == t["recorded_value_evals"] and t["n_policy_evals"] == t["recorded_policy_evals"]
    session = json.load(open(os.path.join(GOLDEN, "gtp_transcript.json")))["session"]
    assert d["gtp"]["all_equal"] and d["gtp"]["mismatches"] == [] and d["gtp"]["commands"] == len(session)
    # what the reference used of the network objects: exactly the duck-typed surface SURVEY 8b lists
    calls = d["surface_calls"]
    for cls in ("HipPolicyNet", "HipValueNet"):
        assert calls[f"{cls}.load_state_dict"] == 1 and calls[f"{cls}.eval"] == 1 and calls[f"{cls}.to"] == 3 and calls[f"{cls}.__call__"] > 100
    assert set(d["input_shapes_seen"]) == {"logits(1, 27, 9, 9)", "value(1, 27, 9, 9)"}   # one position per call (nnet.py:272,283)
