#!/usr/bin/env python3
"""tests/golden/hard_positions.npz: the reference's outputs on the positions where the two GPU kernels disagree
most.  tools/precision_sweep.py (GPU box) evaluates 49,152 seeded random-playout positions with the f16x2 and
the exact-fp32 kernel and keeps the 48 with the largest logit difference (gpurun_out/precision_worst.npz);
this script runs the REFERENCE's PolicyNet / ValueNet (torch, fp32, and float64 as ground truth) on those
positions and stores features + outputs.  Needs the reference checkout (env BOKEGO_REFERENCE); only the
resulting data travels.

    python tools/gen_hard_golden.py [gpurun_out/precision_worst.npz]
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv, argv = sys.argv[:1], sys.argv[1:]          # gen_golden parses its own flags
sys.path.insert(0, os.path.join(REPO, "tools"))
import gen_golden as G  # noqa: E402  (imports the reference, refuses to run without it)
import torch  # noqa: E402

src = argv[0] if argv else os.path.join(REPO, "gpurun_out", "precision_worst.npz")
d = np.load(src)
x = torch.from_numpy(d["feats"].astype(np.float32))
pi, v = G.build_nets()
with torch.no_grad():
    logits, values = pi(x).numpy(), v(x).numpy().reshape(-1)
    logits64, values64 = pi.double()(x.double()).numpy(), v.double()(x.double()).numpy().reshape(-1)
out = os.path.join(REPO, "tests", "golden", "hard_positions.npz")
np.savez_compressed(out, features=d["feats"].astype(np.uint8), logits=logits.astype(np.float32), values=values.astype(np.float32),
                    logits_f64=logits64, values_f64=values64)
rep = {"positions": len(x), "max_abs_logit": float(np.abs(logits64).max())}
for name, lg, va in (("f16x2 kernel", d["logits_f16x2"], d["value_f16x2"]), ("fp32 kernel", d["logits_f32"], d["value_f32"]),
                     ("reference (torch fp32)", logits, values)):
    rep[name] = {"dlogit_vs_reference": float(np.abs(lg - logits).max()), "dlogit_vs_float64": float(np.abs(lg - logits64).max()),
                 "dvalue_vs_reference": float(np.abs(va - values).max()), "dvalue_vs_float64": float(np.abs(va - values64).max())}
import json  # noqa: E402
print(json.dumps(rep, indent=1))
