"""Soak of the small-batch (cooperative) launches: random request sizes 1..127 with random policy prefixes, every output bit
compared with what the big-batch launch gave for the same positions (results do not depend on how a request is launched).
    python tools/coop_soak.py [requests] [seed] [engines]
engines = 2: two engines (two streams) issue their requests at the same time, so cooperative kernels of both compete for the
CUs -- workgroups that cannot meet their peers in time give up and the request is redone (bk_stats().coop_fallbacks): same bits."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402,F401
from bench import make_workload  # noqa: E402
from bokego_amd.bkw import load_bkw  # noqa: E402
from bokego_amd.engine import LeafEngine  # noqa: E402

n_req = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
n_eng = int(sys.argv[3]) if len(sys.argv) > 3 else 1
engs = [LeafEngine(load_bkw(os.path.join(g, "policy_19.bkw")), load_bkw(os.path.join(g, "value_synth.bkw")), device_id=0, max_batch=4096)
        for _ in range(n_eng)]
eng = engs[0]
planes, recs = make_workload(4096, 0)
u8 = planes.astype(np.uint8)
ref = eng.eval(u8, logits=True, probs=True, value=True)          # 3-board workgroups
recs = np.ascontiguousarray(recs)
bad = 0
t0 = time.perf_counter()
inflight = []
for i in range(n_req):
    B = int(rng.integers(1, 128))
    npol = int(rng.choice([0, 1, 1, 1, B, int(rng.integers(0, B + 1))]))
    idx = rng.integers(0, 4096, size=B)
    kind = i % 3
    eng = engs[(i // 3) % n_eng]
    if kind == 0 and recs is not None:
        t = eng.submit_positions(recs[idx], logits=npol > 0, probs=npol > 0, value=True, n_policy=npol)
    elif kind == 1:
        t = eng.submit(u8[idx], logits=npol > 0, probs=npol > 0, value=True, n_policy=npol)
    else:
        t = eng.submit(planes[idx], logits=npol > 0, probs=npol > 0, value=True, n_policy=npol)
    inflight.append((eng, t, idx, npol))
    if len(inflight) == 3 * n_eng or i == n_req - 1:
        for we, t, idx, npol in inflight:
            out = we.wait(t)
            ok = np.array_equal(out["value"].view(np.uint32), ref["value"][idx].view(np.uint32))
            if npol:
                ok = ok and np.array_equal(out["logits"].view(np.uint32), ref["logits"][idx[:npol]].view(np.uint32))
                ok = ok and np.array_equal(out["probs"].view(np.uint32), ref["probs"][idx[:npol]].view(np.uint32))
            if not ok:
                bad += 1
                if bad <= 5:
                    print(f"MISMATCH at request {i}: B={len(idx)} n_policy={npol}", flush=True)
        inflight = []
    if i % 10000 == 9999:
        print(f"{i + 1} requests, {bad} mismatches, {time.perf_counter() - t0:.0f} s", flush=True)
st = [e.stats() for e in engs]
print(f"{n_req} requests on {n_eng} engine(s): {bad} mismatches; cooperative launches {sum(s['coop_launches'] for s in st)}, "
      f"fallbacks {sum(s['coop_fallbacks'] for s in st)}, {time.perf_counter() - t0:.0f} s")
sys.exit(1 if bad else 0)
