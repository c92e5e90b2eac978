"""Developer script for the GPU box: parity numbers + a rough per-batch-size timing."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bokego_amd.bkw import load_bkw  # noqa: E402
from bokego_amd.engine import LeafEngine  # noqa: E402

G = os.path.join(REPO, "tests", "golden")
pw, vw = load_bkw(f"{G}/policy_19.bkw"), load_bkw(f"{G}/value_synth.bkw")
f = np.load(f"{G}/features.npz")["incremental"].astype(np.float32)
n = np.load(f"{G}/nets.npz")
e = LeafEngine(pw, vw, max_batch=4096)
for B in (1, 2, 3, 81, 536):
    o = e.eval(f[:B], logits=True, probs=True, value=True)
    print(B, "dlogit %.3g dprob %.3g dvalue %.3g" % (np.abs(o["logits"] - n["logits_b1"][:B]).max(),
          np.abs(o["probs"] - n["probs_b1"][:B]).max(), np.abs(o["value"] - n["values_b1"][:B]).max()), flush=True)
import torch  # noqa: E402

x = torch.from_numpy(f[np.random.default_rng(0).integers(0, len(f), 4096)]).cuda()
e.set_profiling(True)
for B in (1, 64, 256, 768, 1536, 4096):
    for _ in range(3):
        e.eval_device(x[:B], logits=True, probs=True, value=True)
    torch.cuda.synchronize()
    s0 = e.stats()
    t0 = time.time()
    for _ in range(10):
        e.eval_device(x[:B], logits=True, probs=True, value=True)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 10
    s1 = e.stats()
    kms = (s1["kernel_ms_sum"] - s0["kernel_ms_sum"]) / (s1["kernel_ms_count"] - s0["kernel_ms_count"])
    print(f"B={B}: wall {dt*1e3:.3f} ms  kernel {kms:.3f} ms  {B/kms*1e3:.0f} leaf/s  "
          f"{B/kms*1e3*266.838e6/157.3e12*100:.1f}% of fp32 MFMA roof", flush=True)
