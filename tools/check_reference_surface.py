#!/usr/bin/env python3
"""Reference-side drop-in check (build container only).

INTEGRATION.md section 1 claims that the reference's own `MCTS` / `GTP` (bokego/mcts.py:46-79, bokego/gtp.py:47-55)
run UNCHANGED on the shim networks of bokego_amd/nnet.py -- that swapping `PolicyNet()` / `ValueNet()` for
`HipPolicyNet()` / `HipValueNet()` in boke.py:30-38 is the whole integration.  This script checks the reference side of
that claim: it imports the reference (env BOKEGO_REFERENCE, default /root/reference), builds the two shims exactly the
way boke.py builds its nets (construct, load_state_dict of the shipped .pt-shaped state dicts, eval()), hands them to the
REFERENCE's MCTS and GTP classes, and replays
    * the recorded search traces (tests/golden/mcts_trace.json: r300_t20 and r1600, made by the reference on its own
      torch nets): chosen moves and every root child's visit count must come out the same;
    * the recorded GTP session (tests/golden/gtp_transcript.json): every reply must be the same text.
There is no GPU in the build container, so the shims' single engine hook `_HipNet._run` is served by the CPU oracle
(oracle/oracle.py) here -- everything above that hook (`__call__`, `.to`, `.eval`, `.load_state_dict`, `.state_dict`,
tensor shapes and dtypes handed to nnet.policy_dist / nnet.value, nnet.py:265-284) is the product's code.  The same
shims on the real engine are covered on the GPU box by tests/test_gpu_mcts.py (which cannot import the reference).

Writes tests/golden/reference_surface.json (data only: what was called, what matched); tests/test_reference_surface.py
asserts on that file.  Nothing of the reference travels.

    python tools/check_reference_surface.py [--skip-r1600]
"""
import argparse
import collections
import hashlib
import json
import os
import random
import sys
import time

REF = os.environ.get("BOKEGO_REFERENCE", "/root/reference")
if not os.path.isdir(os.path.join(REF, "bokego")):
    sys.exit(f"reference checkout not found at {REF}; set BOKEGO_REFERENCE")

random.seed(0)            # before importing bokego.go: its Zobrist table is drawn at import (go.py:48-49), as in gen_golden.py
sys.path.insert(0, REF)
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.set_grad_enabled(False)      # boke.py:11

import bokego.gtp as rgtp  # noqa: E402
import bokego.mcts as rmcts  # noqa: E402

from bokego_amd import nnet as shim  # noqa: E402
from bokego_amd.bkw import load_bkw  # noqa: E402
from oracle.oracle import OraclePolicy, OracleValue  # noqa: E402

GOLDEN = os.path.join(REPO, "tests", "golden")
CALLS = collections.Counter()
SHAPES = collections.Counter()


def _counted(cls, oracle_cls, key):
    """The product's shim class with its engine hook served by the CPU oracle and its surface instrumented."""

    class Shim(cls):
        def _run(self, x, **want):
            self._need_weights()
            if getattr(self, "_oracle", None) is None:
                self._oracle = oracle_cls(self._sd)
            assert isinstance(x, torch.Tensor) and x.dtype == torch.float32 and tuple(x.shape[1:]) == (27, 9, 9), (type(x), getattr(x, "shape", None))
            SHAPES[f"{key}{tuple(x.shape)}"] += 1
            return {key: torch.from_numpy(np.asarray(self._oracle(x.numpy())))}, None

        def _drop_engine(self):
            self._oracle = None

        def __call__(self, x):
            CALLS[f"{cls.__name__}.__call__"] += 1
            return super().__call__(x)

        def to(self, device):
            CALLS[f"{cls.__name__}.to"] += 1
            return super().to(device)

        def eval(self):
            CALLS[f"{cls.__name__}.eval"] += 1
            return super().eval()

        def load_state_dict(self, sd, strict=True):
            CALLS[f"{cls.__name__}.load_state_dict"] += 1
            return super().load_state_dict(sd, strict)

    Shim.__name__ = cls.__name__
    return Shim


def build_nets():
    """boke.py:30-38 with the shim classes: Net(); load_state_dict(checkpoint["model_state_dict"]); eval()."""
    P, V = _counted(shim.HipPolicyNet, OraclePolicy, "logits"), _counted(shim.HipValueNet, OracleValue, "value")
    ck_p = {"model_state_dict": {k: torch.from_numpy(v) for k, v in load_bkw(os.path.join(GOLDEN, "policy_19.bkw")).items()}}
    ck_v = {"model_state_dict": {k: torch.from_numpy(v) for k, v in load_bkw(os.path.join(GOLDEN, "value_synth.bkw")).items()}}
    pi = P()
    pi.load_state_dict(ck_p["model_state_dict"])
    pi.eval()
    val = V()
    val.load_state_dict(ck_v["model_state_dict"])
    val.eval()
    return pi, val


def clear_reference_caches():
    rmcts.MCTS._val_cache.clear(); rmcts.MCTS._dist_cache.clear(); rmcts.MCTS._fts_cache.clear()


def replay_trace(pi, val, rec):
    clear_reference_caches()
    torch.manual_seed(0)
    tree = rmcts.MCTS(rmcts.Go_MCTS(), pi, val, no_sim=True, **rec["kwargs"])       # the REFERENCE's search
    out, t0 = [], time.time()
    for want in rec["moves"]:
        tree.rollout(rec["rollouts"])
        root = tree.root
        kids = {str(int(c.last_move)): int(tree.N[c]) for c in tree.children[root]}
        wr = float(tree.winrate())
        best = tree.choose()
        out.append({"move": int(best.last_move), "move_equal": int(best.last_move) == want["move"],
                    "child_N_equal": kids == {k: int(v) for k, v in want["child_N"].items()},
                    "root_winrate_delta": abs(wr - want["root_winrate"])})
    return {"rollouts": rec["rollouts"], "kwargs": rec["kwargs"], "moves": out, "seconds": time.time() - t0,
            "all_equal": all(m["move_equal"] and m["child_N_equal"] for m in out),
            "n_value_evals": len(rmcts.MCTS._val_cache), "n_policy_evals": len(rmcts.MCTS._dist_cache),
            "recorded_value_evals": rec["n_value_evals"], "recorded_policy_evals": rec["n_policy_evals"]}


def replay_gtp(pi, val, t):
    clear_reference_caches()
    torch.manual_seed(0)
    g = rgtp.GTP(rmcts.Go_MCTS(), pi, val, no_sim=True, time_lim=None, n_rollouts=t["n_rollouts"], pondering=False)   # the REFERENCE's GTP
    g.running = True
    bad = []
    for cmd, want in t["session"]:
        got = g.send(cmd)
        if got != want:
            bad.append({"cmd": cmd, "want": want, "got": got})
    return {"commands": len(t["session"]), "mismatches": bad, "all_equal": not bad}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-r1600", action="store_true")
    args = ap.parse_args()
    pi, val = build_nets()
    traces = json.load(open(os.path.join(GOLDEN, "mcts_trace.json")))
    res = {"what": "the reference's MCTS / GTP classes run on bokego_amd.nnet.HipPolicyNet / HipValueNet (engine hook served "
                   "by the CPU oracle), replaying the reference's own recorded traces and GTP session",
           "reference_classes": [f"{rmcts.MCTS.__module__}.{rmcts.MCTS.__qualname__}", f"{rgtp.GTP.__module__}.{rgtp.GTP.__qualname__}"],
           "shim_classes": ["bokego_amd.nnet.HipPolicyNet", "bokego_amd.nnet.HipValueNet"],
           "shim_sha256": hashlib.sha256(open(os.path.join(REPO, "bokego_amd", "nnet.py"), "rb").read()).hexdigest(),
           "traces": {}}
    for name in ("r300_t20",) + (() if args.skip_r1600 else ("r1600",)):
        res["traces"][name] = replay_trace(pi, val, traces[name])
        print(name, "equal" if res["traces"][name]["all_equal"] else "DIFFERENT", f"{res['traces'][name]['seconds']:.1f}s")
    res["gtp"] = replay_gtp(pi, val, json.load(open(os.path.join(GOLDEN, "gtp_transcript.json"))))
    print("gtp", "equal" if res["gtp"]["all_equal"] else res["gtp"]["mismatches"][:3])
    res["surface_calls"] = dict(sorted(CALLS.items()))
    res["input_shapes_seen"] = dict(sorted(SHAPES.items()))
    with open(os.path.join(GOLDEN, "reference_surface.json"), "w") as f:
        json.dump(res, f, indent=1)
    ok = all(t["all_equal"] for t in res["traces"].values()) and res["gtp"]["all_equal"]
    print("reference surface:", "OK" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
