"""Round-trip latency of small host-buffer evaluations (the single-tree genmove regime): B = 1, 8, 82."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from bokego_amd.bkw import load_bkw
from bokego_amd.engine import LeafEngine
from bokego_amd.workload import make_batch
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
eng = LeafEngine(load_bkw(os.path.join(g, "policy_19.bkw")), load_bkw(os.path.join(g, "value_synth.bkw")), max_batch=128)
x, recs = make_batch(82, dtype=np.uint8, with_records=True)
eng.set_profiling(True)
for B, npol in ((1, 1), (8, 1), (82, 1), (82, 82)):
    for fn, arg, name in ((eng.submit, x, "u8 planes"), (eng.submit_positions, recs, "records")):
        for _ in range(20):
            eng.wait(fn(arg[:B], logits=False, probs=True, value=True, n_policy=npol))
        s0 = eng.stats()
        t = time.perf_counter()
        for _ in range(200):
            eng.wait(fn(arg[:B], logits=False, probs=True, value=True, n_policy=npol))
        dt = (time.perf_counter() - t) / 200 * 1e6
        s1 = eng.stats()
        k = (s1["kernel_ms_sum"] - s0["kernel_ms_sum"]) / (s1["kernel_ms_count"] - s0["kernel_ms_count"]) * 1e3
        print(f"B={B:3d} n_policy={npol:2d} {name:10s}: round trip {dt:7.1f} us, leaf kernel {k:6.1f} us")
