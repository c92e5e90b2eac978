"""Kernel time of small batches: the one-CU-per-board form against the cooperative form with 2 .. 12 CUs per board.
    python tools/coop_probe.py          (BK_COOP=0|2|4|8 in the environment forces one form)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bokego_amd.bkw import load_bkw  # noqa: E402
from bokego_amd.engine import LeafEngine  # noqa: E402
from bokego_amd.workload import make_batch  # noqa: E402

g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
eng = LeafEngine(load_bkw(os.path.join(g, "policy_19.bkw")), load_bkw(os.path.join(g, "value_synth.bkw")), max_batch=512)
x = make_batch(512, seed_base=1, dtype=np.uint8)
eng.set_profiling(True)
print("BK_COOP =", os.environ.get("BK_COOP", "(default)"))
for B in (1, 8, 16, 31, 32, 48, 62, 63, 64, 70, 84, 85, 100, 127, 200):
    for _ in range(3):
        eng.eval(x[:B], probs=True, value=True, n_policy=1)
    s0 = eng.stats()
    for _ in range(20):
        eng.eval(x[:B], probs=True, value=True, n_policy=1)
    s1 = eng.stats()
    n = s1["kernel_ms_count"] - s0["kernel_ms_count"]
    print(f"B {B:4d} (+1 policy): {1e3 * (s1['kernel_ms_sum'] - s0['kernel_ms_sum']) / n:8.1f} us per call   "
          f"coop launches {s1['coop_launches'] - s0['coop_launches']}  fallbacks {s1['coop_fallbacks'] - s0['coop_fallbacks']}")
