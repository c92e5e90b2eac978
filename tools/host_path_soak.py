"""Soak of the host-buffer entry points at every size: random B in 1..8192 (small cooperative launches, one-round launches,
split launches, the copy pool, the two-part launch of big f32 requests), 1-4 tickets in flight, f32 / u8 planes and position
records, both precisions; every output bit compared with the same engine's device-resident evaluation of the same positions.
    python tools/host_path_soak.py [requests] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from bench import make_workload  # noqa: E402
from bokego_amd.bkw import load_bkw  # noqa: E402
from bokego_amd.engine import LeafEngine  # noqa: E402

n_req = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
N = 8192
planes, recs = make_workload(N, 0)
u8 = planes.astype(np.uint8)
bad = total = 0
t0 = time.perf_counter()
for prec in ("f32", "f16x2"):
    eng = LeafEngine(load_bkw(os.path.join(g, "policy_19.bkw")), load_bkw(os.path.join(g, "value_synth.bkw")), device_id=0, max_batch=N, precision=prec)
    d = eng.eval_device(torch.from_numpy(u8).cuda(), logits=True, probs=True, value=True)
    ref = {k: d[k].cpu().numpy() for k in ("logits", "probs", "value")}
    inflight = []
    for i in range(n_req):
        r = rng.random()
        B = int(rng.integers(1, 128)) if r < 0.3 else int(rng.integers(128, 1024)) if r < 0.6 else int(rng.integers(1024, N + 1))
        npol = int(rng.choice([0, 1, B, int(rng.integers(0, B + 1))]))
        lo = int(rng.integers(0, N - B + 1))
        idx = np.arange(lo, lo + B) if rng.random() < 0.5 else rng.integers(0, N, size=B)
        kind = int(rng.integers(0, 3))
        src = (recs, u8, planes)[kind][idx]
        if kind == 0:
            t = eng.submit_positions(src, logits=npol > 0, probs=npol > 0, value=True, n_policy=npol)
        else:
            t = eng.submit(src, logits=npol > 0, probs=npol > 0, value=True, n_policy=npol)
        inflight.append((t, idx, npol, kind))
        depth = int(rng.integers(1, 5))
        while len(inflight) >= depth or (i == n_req - 1 and inflight):
            t, idx, npol, kind = inflight.pop(0)
            out = eng.wait(t)
            ok = np.array_equal(out["value"].view(np.uint32), ref["value"][idx].view(np.uint32))
            if npol:
                ok = ok and np.array_equal(out["logits"].view(np.uint32), ref["logits"][idx[:npol]].view(np.uint32))
                ok = ok and np.array_equal(out["probs"].view(np.uint32), ref["probs"][idx[:npol]].view(np.uint32))
            total += 1
            if not ok:
                bad += 1
                if bad <= 8:
                    print(f"MISMATCH {prec}: B={len(idx)} n_policy={npol} kind={('positions', 'u8', 'f32')[kind]}", flush=True)
        if i % 500 == 499:
            print(f"{prec}: {i + 1} requests, {bad} mismatches, {time.perf_counter() - t0:.0f} s", flush=True)
    s = eng.stats()
    print(f"{prec}: f16 overflow redos {s['f16_overflow_fallbacks']}, cooperative launches {s['coop_launches']} (fallbacks {s['coop_fallbacks']}), split launches {s['split_launches']}")
    eng.close()
print(f"{total} requests checked: {bad} mismatches, {time.perf_counter() - t0:.0f} s")
sys.exit(1 if bad else 0)
