"""One request far beyond the benchmark's batch: B = 65,536 positions (143 MB of u8 planes, 43,692 workgroups) through the
host-buffer and the device-resident entry points, both precisions, bit for bit against the same positions evaluated 4,096 at a time.
    python tools/big_batch_check.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import make_workload  # noqa: E402
from bokego_amd.bkw import load_bkw  # noqa: E402
from bokego_amd.engine import LeafEngine  # noqa: E402

g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
planes, _ = make_workload(4096, 0)
u8 = np.tile(planes.astype(np.uint8), (16, 1, 1, 1))
perm = np.random.default_rng(0).permutation(len(u8))
u8 = u8[perm]
ok = True
for prec in ("f32", "f16x2"):
    eng = LeafEngine(load_bkw(os.path.join(g, "policy_19.bkw")), load_bkw(os.path.join(g, "value_synth.bkw")), device_id=0, max_batch=65536, precision=prec)
    ref = [eng.eval(u8[i:i + 4096], logits=True, probs=True, value=True) for i in range(0, len(u8), 4096)]
    ref = {k: np.concatenate([r[k] for r in ref]) for k in ("logits", "probs", "value")}
    t = time.perf_counter()
    big = eng.eval(u8, logits=True, probs=True, value=True)
    dt = time.perf_counter() - t
    d = eng.eval_device(torch.from_numpy(u8).cuda(), logits=True, probs=True, value=True)
    torch.cuda.synchronize()
    for k in ref:
        a = np.array_equal(big[k].view(np.uint32), ref[k].view(np.uint32))
        b = np.array_equal(d[k].cpu().numpy().view(np.uint32), ref[k].view(np.uint32))
        ok = ok and a and b
        print(f"{prec} {k}: host path {'==' if a else '!='} chunks, device path {'==' if b else '!='} chunks")
    print(f"{prec}: B = {len(u8)} in {dt * 1e3:.1f} ms = {len(u8) / dt:,.0f} leaf-evals/s through the host path")
    eng.close()
sys.exit(0 if ok else 1)
