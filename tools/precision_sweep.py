"""f16x2 kernel vs exact fp32 kernel on many seeded random-playout positions: largest output differences and
whether the fp16 range guard ever fires.  usage (GPU box): python tools/precision_sweep.py [n_batches]"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from bokego_amd.bkw import load_bkw
from bokego_amd.engine import LeafEngine
from bokego_amd.workload import make_batch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
pw, vw = load_bkw(os.path.join(g, "policy_19.bkw")), load_bkw(os.path.join(g, "value_synth.bkw"))
a = LeafEngine(pw, vw, max_batch=4096, precision="f16x2")
b = LeafEngine(pw, vw, max_batch=4096, precision="f32")
worst = dict(logit=0.0, prob=0.0, value=0.0, absmax_logit=0.0)
keep = []   # (diff, features u8, f16x2 logits, fp32 logits, f16x2 value, fp32 value) of the worst positions
for i in range(n):
    x = make_batch(4096, seed_base=1_000_000 + i * 4096, dtype=np.uint8)
    oa = a.eval(x, logits=True, probs=True, value=True)
    ob = b.eval(x, logits=True, probs=True, value=True)
    worst["logit"] = max(worst["logit"], float(np.abs(oa["logits"] - ob["logits"]).max()))
    worst["prob"] = max(worst["prob"], float(np.abs(oa["probs"] - ob["probs"]).max()))
    worst["value"] = max(worst["value"], float(np.abs(oa["value"] - ob["value"]).max()))
    worst["absmax_logit"] = max(worst["absmax_logit"], float(np.abs(ob["logits"]).max()))
    d = np.abs(oa["logits"] - ob["logits"]).max(axis=1)
    for j in np.argsort(d)[-8:]:
        keep.append((float(d[j]), x[j].copy(), oa["logits"][j].copy(), ob["logits"][j].copy(), float(oa["value"][j]), float(ob["value"][j])))
    keep = sorted(keep, key=lambda t: -t[0])[:48]
    print(i, worst, flush=True)
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "precision_worst.npz")
os.makedirs(os.path.dirname(out), exist_ok=True)
np.savez_compressed(out, diff=np.array([k[0] for k in keep]), feats=np.stack([k[1] for k in keep]),
                    logits_f16x2=np.stack([k[2] for k in keep]), logits_f32=np.stack([k[3] for k in keep]),
                    value_f16x2=np.array([k[4] for k in keep]), value_f32=np.array([k[5] for k in keep]))
st = a.stats()
print(json.dumps({"positions": n * 4096, **worst, "f16_overflow_fallbacks": st["f16_overflow_fallbacks"]}))
