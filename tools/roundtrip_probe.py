"""Round trip of a small request through the host-buffer ABI (what a one-tree genmove pays per expansion batch) against the leaf
kernel's own time (HIP events): from u8 planes (H2D copy of the planes) and from 192-byte position records (copy-free path:
encoder kernel + leaf kernel).  Round 4, one MI355X: 63 tasks 122 us round trip for a 102-104 us kernel, 80 tasks 153 for 134:
~20 us around the kernel (launch latency of two kernels, the 5 us encoder, the host noticing the event).
    python tools/roundtrip_probe.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from bokego_amd.bkw import load_bkw
from bokego_amd.engine import LeafEngine
from bokego_amd.workload import make_batch
g = os.path.join(os.getcwd(), "tests", "golden")
eng = LeafEngine(load_bkw(os.path.join(g, "policy_19.bkw")), load_bkw(os.path.join(g, "value_synth.bkw")), max_batch=256)
x8, recs = make_batch(128, seed_base=3, dtype=np.uint8, with_records=True)
for B in (63, 79):
    for name, f in (("planes u8 (no encoder kernel)", lambda: eng.eval(x8[:B], probs=True, value=True, n_policy=1)),
                    ("position records (encoder kernel + leaf kernel)", lambda: eng.wait(eng.submit_positions(recs[:B], probs=True, value=True, n_policy=1)))):
        for _ in range(50): f()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(400): f()
            ts.append((time.perf_counter() - t0) / 400 * 1e6)
        eng.set_profiling(True); s0 = eng.stats()
        for _ in range(100): f()
        s1 = eng.stats(); eng.set_profiling(False)
        k = 1e3 * (s1["kernel_ms_sum"] - s0["kernel_ms_sum"]) / (s1["kernel_ms_count"] - s0["kernel_ms_count"])
        print(f"B {B} {name}: round trip {min(ts):.1f} us (median {sorted(ts)[2]:.1f}), leaf kernel {k:.1f} us")
