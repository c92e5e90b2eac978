#!/usr/bin/env python3
"""tests/golden/sweep_worst.npz + sweep_summary.json: the worst cases of the 49,152-position x 2-weight-set sweep
AGAINST THE REFERENCE, as committed fixtures.

Inputs: gpurun_out/sweep_worst.npz + gpurun_out/sweep_summary.json from tools/sweep_vs_reference.py (GPU box: for
each weight set and each kernel the positions with the largest |dlogit| / |dvalue| / |dprob| against the
reference), and tests/golden/_sweep/ref_{A,B}.npz from tools/gen_sweep_reference.py (the reference's outputs on
the whole sweep).  Output per weight set: features u8, the reference's fp32 logits / values and their float64
ground truth.  Data only; no reference code is involved here.
"""
import json
import os
import shutil

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(REPO, "tests", "golden")
w = np.load(os.path.join(REPO, "gpurun_out", "sweep_worst.npz"))
out = {}
for s in ("A", "B"):
    ref = np.load(os.path.join(G, "_sweep", f"ref_{s}.npz"))
    idx = w[f"index_{s}"]
    out[f"index_{s}"] = idx
    out[f"seed_{s}"] = w[f"seed_{s}"]
    out[f"features_{s}"] = w[f"features_{s}"].astype(np.uint8)
    out[f"logits_{s}"] = ref["logits"][idx]
    out[f"values_{s}"] = ref["values"][idx]
    out[f"logits_f64_{s}"] = ref["logits"][idx].astype(np.float64) + ref["dlogits64"][idx]
    out[f"values_f64_{s}"] = ref["values"][idx].astype(np.float64) + ref["dvalues64"][idx]
    print(s, len(idx), "positions, max |logit|", float(np.abs(out[f"logits_{s}"]).max()))
np.savez_compressed(os.path.join(G, "sweep_worst.npz"), **out)
shutil.copy(os.path.join(REPO, "gpurun_out", "sweep_summary.json"), os.path.join(G, "sweep_summary.json"))
print(json.dumps(json.load(open(os.path.join(G, "sweep_summary.json")))["A"]["f32"]["dlogit_vs_reference"]))
