#!/usr/bin/env python3
"""GPU box: both kernels against the REFERENCE on the whole precision sweep.

Inputs: tests/golden/_sweep/ref_{A,B}.npz written by tools/gen_sweep_reference.py in the build container (the
reference's fp32 outputs and their float64 ground truth on 49,152 seeded random-playout positions, two weight
sets).  The positions are regenerated here from their seeds by the build's own board engine.

Outputs (gpurun_out/): sweep_summary.json -- per weight set and precision: max / p99.9 |dlogit|, |dprob|, |dvalue|
against the reference's fp32 AND against float64, positions over 1e-4, overflow fallbacks -- and sweep_worst.npz,
the worst-k positions per (set, precision, output) with the kernels' outputs, for tools/gen_sweep_golden.py.

    python tools/sweep_vs_reference.py [k=24]
"""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bokego_amd.bkw import load_bkw  # noqa: E402
from bokego_amd.engine import LeafEngine  # noqa: E402
from bokego_amd.workload import make_batch  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 24
G = os.path.join(REPO, "tests", "golden")


def weight_sets():
    """set A = the goldens' nets; set B = the two trained trunks swapped + another seeded value head
    (tools/gen_sweep_reference.py:build_nets_b; value_synth.bkw carries every policy_17 tensor)."""
    p19, vs = load_bkw(os.path.join(G, "policy_19.bkw")), load_bkw(os.path.join(G, "value_synth.bkw"))
    head_b = np.load(os.path.join(G, "value_head_b.npz"))
    pol_b = {k: v for k, v in vs.items() if k.startswith("conv.")}
    val_b = dict(p19)
    val_b.update({k: head_b[k] for k in head_b.files})
    return {"A": (p19, vs), "B": (pol_b, val_b)}


def softmax(lg):
    e = np.exp(lg - lg.max(1, keepdims=True))
    return e / e.sum(1, keepdims=True)


def main():
    summary, worst = {}, {}
    sets = weight_sets()
    refs = {s: np.load(os.path.join(G, "_sweep", f"ref_{s}.npz")) for s in sets}
    n = int(refs["A"]["n"])
    seed0 = int(refs["A"]["seed0"])
    feats = np.concatenate([make_batch(4096, seed_base=seed0 + i, dtype=np.uint8) for i in range(0, n, 4096)])
    print("positions ready", feats.shape, flush=True)
    for s, (pw, vw) in sets.items():
        ref = refs[s]
        lg_ref, va_ref = ref["logits"], ref["values"]
        lg64, va64 = lg_ref.astype(np.float64) + ref["dlogits64"], va_ref.astype(np.float64) + ref["dvalues64"]
        pr_ref, pr64 = softmax(lg_ref.astype(np.float64)), softmax(lg64)
        rep = {"positions": n, "max_abs_logit": float(np.abs(lg_ref).max()),
               "reference_fp32_vs_float64": {"dlogit_max": float(np.abs(ref["dlogits64"]).max()),
                                             "dvalue_max": float(np.abs(ref["dvalues64"]).max())}}
        for prec in ("f32", "f16x2"):
            eng = LeafEngine(pw, vw, max_batch=4096, precision=prec)
            outs = [eng.eval(feats[i:i + 4096], logits=True, probs=True, value=True) for i in range(0, n, 4096)]
            st = eng.stats()
            eng.close()
            lg = np.concatenate([o["logits"] for o in outs])
            pr = np.concatenate([o["probs"] for o in outs])
            va = np.concatenate([o["value"] for o in outs])
            dl, dl64 = np.abs(lg - lg_ref).max(1), np.abs(lg - lg64).max(1)
            dp, dv, dv64 = np.abs(pr - pr_ref).max(1), np.abs(va - va_ref), np.abs(va - va64)
            rep[prec] = {
                "dlogit_vs_reference": {"max": float(dl.max()), "p999": float(np.quantile(dl, 0.999)), "mean": float(dl.mean()),
                                        "positions_over_1e-4": int((dl > 1e-4).sum())},
                "dlogit_vs_float64": {"max": float(dl64.max()), "p999": float(np.quantile(dl64, 0.999)), "mean": float(dl64.mean())},
                "dprob_vs_reference": {"max": float(dp.max()), "positions_over_1e-5": int((dp > 1e-5).sum())},
                "dprob_vs_float64_max": float(np.abs(pr - pr64).max()),
                "dvalue_vs_reference": {"max": float(dv.max()), "positions_over_1e-4": int((dv > 1e-4).sum())},
                "dvalue_vs_float64_max": float(dv64.max()),
                "argmax_mismatches_vs_reference": int((lg.argmax(1) != lg_ref.argmax(1)).sum()),
                "f16_overflow_fallbacks": int(st["f16_overflow_fallbacks"]),
            }
            idx = np.unique(np.concatenate([np.argsort(dl)[-K:], np.argsort(dv)[-K // 2:], np.argsort(dp)[-K // 2:]]))
            worst[(s, prec)] = idx
            print(s, prec, json.dumps(rep[prec]), flush=True)
        summary[s] = rep
    out = os.path.join(REPO, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(summary, open(os.path.join(out, "sweep_summary.json"), "w"), indent=1)
    pack = {}
    for s in sets:
        idx = np.unique(np.concatenate([worst[(s, p)] for p in ("f32", "f16x2")]))
        pack[f"index_{s}"] = idx
        pack[f"seed_{s}"] = seed0 + idx
        pack[f"features_{s}"] = feats[idx]
    np.savez_compressed(os.path.join(out, "sweep_worst.npz"), **pack)
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
