"""Where a 1600-rollout genmove spends its time on the native tree: evaluator calls, their batch sizes and latency.
    python tools/genmove_probe.py [moves] [precision]"""
import os
import sys
import time
from collections import Counter

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402,F401

from bokego_amd import mcts, nnet  # noqa: E402
from bokego_amd.bkw import load_bkw  # noqa: E402
from bokego_amd.mcts_native import NativeMCTS  # noqa: E402

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
moves = int(sys.argv[1]) if len(sys.argv) > 1 else 30
precision = sys.argv[2] if len(sys.argv) > 2 else None
pi, val = nnet.HipPolicyNet(precision=precision), nnet.HipValueNet(precision=precision)
pi.load_state_dict(load_bkw(os.path.join(G, "policy_19.bkw")))
val.load_state_dict(load_bkw(os.path.join(G, "value_synth.bkw")))
kw = {}          # BK_SPECULATE / BK_SPECULATE_ROWS override NativeMCTS's defaults (N = 50 with 256 rows on f16x2 engines, 80 rows on fp32 ones)
if "BK_SPECULATE" in os.environ:
    kw["speculate"] = int(os.environ["BK_SPECULATE"])
if "BK_SPECULATE_ROWS" in os.environ:
    kw["speculate_rows"] = int(os.environ["BK_SPECULATE_ROWS"])
if "BK_REQUEST_TASKS" in os.environ:       # 0: no request-size steps (round-2 behaviour); default on fp32 engines: 64 (steps 64 / 80 / 128)
    kw["request_tasks"] = int(os.environ["BK_REQUEST_TASKS"])
if "BK_EAGER_TOP" in os.environ:           # children evaluated at an expansion (0: all)
    kw["eager_top"] = int(os.environ["BK_EAGER_TOP"])
if "BK_REQUEST_STEPS" in os.environ:       # e.g. "128,256,512"
    kw["request_steps"] = tuple(int(v) for v in os.environ["BK_REQUEST_STEPS"].split(","))
tree = NativeMCTS(mcts.Go_MCTS(), pi, val, **kw)
ev = tree.evaluator
sizes, lat = Counter(), []
orig = ev.__class__.__call__


def timed(self, feats, npol):
    t = time.perf_counter()
    r = orig(self, feats, npol)
    lat.append(time.perf_counter() - t)
    sizes[(len(feats), int(npol))] += 1
    return r


ev.__class__.__call__ = timed
tree.rollout(200); tree.choose()       # warm
sizes.clear(); lat.clear()
per_move = []
for _ in range(moves):
    t = time.perf_counter()
    tree.rollout(1600)
    tree.choose()
    per_move.append(time.perf_counter() - t)
    if tree.root._terminal:
        break
n = len(per_move)
print("speculate =", os.environ.get("BK_SPECULATE", "(default)"), " value evals", tree._pool.info(0)["n_value_evals"], " checksum of moves", sum((i + 1) * m for i, m in enumerate(tree._pool.moves(0))))
print(f"moves {n}  ms/move mean {1e3 * sum(per_move) / n:.3f}  min {1e3 * min(per_move):.3f}  max {1e3 * max(per_move):.3f}")
print(f"evaluator calls {len(lat)} = {len(lat) / n:.1f} per move; in-evaluator time {1e3 * sum(lat) / n:.3f} ms/move; "
      f"per call median {1e6 * sorted(lat)[len(lat) // 2]:.0f} us  p10 {1e6 * sorted(lat)[len(lat) // 10]:.0f} us")
import ctypes  # noqa: E402
ph = (ctypes.c_double * 3)()
tree._lib.bk_pool_phase_seconds(tree._pool._h, ph)
print(f"native tree per move (whole run incl. warm-up): advance {1e3 * ph[0] / n:.3f} ms, emit {1e3 * ph[1] / n:.3f} ms, deliver {1e3 * ph[2] / n:.3f} ms; "
      f"outside the evaluator {1e3 * (sum(per_move) - sum(lat)) / n:.3f} ms/move")
print("batch (boards, policy rows) -> calls:", dict(sizes.most_common(12)))
bands = Counter()
for (rows, npol), c in sizes.items():
    tasks = rows + npol
    bands["<=21" if tasks <= 21 else "22-32" if tasks <= 32 else "33-42" if tasks <= 42 else "43-64" if tasks <= 64 else
          "65-85" if tasks <= 85 else "86-128" if tasks <= 128 else ">128"] += c
over = Counter()
for (rows, npol), c in sizes.items():
    if 60 <= rows + npol <= 90:
        over[rows + npol] += c
print("requests of 60..90 tasks, by tasks:", dict(sorted(over.items())))
print("requests by network tasks (the launch forms' ranges: 12 / 8 / 6 / 4 / 3 / 2 / 1 CUs per board):",
      {k: bands[k] for k in ("<=21", "22-32", "33-42", "43-64", "65-85", "86-128", ">128")})
