import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from bokego_amd.bkw import load_bkw
from bokego_amd.engine import LeafEngine
from bokego_amd.workload import make_batch
g = "/root/repo/tests/golden"
eng = LeafEngine(load_bkw(g + "/policy_19.bkw"), load_bkw(g + "/value_synth.bkw"), max_batch=512)
x = make_batch(512, seed_base=1, dtype=np.uint8)
eng.set_profiling(True)
def timed(B, npol=1, reps=20):
    for _ in range(3): eng.eval(x[:B], probs=True, value=True, n_policy=npol)
    s0 = eng.stats()
    for _ in range(reps): eng.eval(x[:B], probs=True, value=True, n_policy=npol)
    s1 = eng.stats()
    return 1e3 * (s1["kernel_ms_sum"] - s0["kernel_ms_sum"]) / (s1["kernel_ms_count"] - s0["kernel_ms_count"])
for B in (1, 3, 7, 8, 9, 11, 13, 15, 16, 19, 20, 23, 31):
    row = []
    for forced in (-1, 12, 8, 6):
        try:
            eng.set_option("coop", forced)
            row.append(f"{'default' if forced < 0 else str(forced) + ' CUs'}: {timed(B):6.1f}")
        except Exception as e:
            row.append(f"{forced}: n/a")
    print(f"B {B:3d} (+1 policy = {B+1} tasks): " + " | ".join(row), flush=True)
