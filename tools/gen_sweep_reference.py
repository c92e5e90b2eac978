#!/usr/bin/env python3
"""The REFERENCE's outputs on the whole precision sweep.

49,152 seeded random-playout positions (bokego_amd.workload.make_batch, seeds 1,000,000 ..), two weight
sets, evaluated by the reference's own PolicyNet / ValueNet (torch CPU, fp32 = the parity target, and
float64 = ground truth):

    set A   policy = policy_19               value = policy_17 trunk + seeded head   (the goldens' nets)
    set B   policy = policy_17               value = policy_19 trunk + another seeded head

Needs the reference checkout (tools/gen_golden.py refuses to import without it).  Output (data only):
    tests/golden/_sweep/ref_{A,B}.npz   logits f32 [N,81], values f32 [N], and float64-minus-fp32 deltas;
                                         git-ignored (64 MB), but they travel to the GPU box with the snapshot
    tests/golden/value_head_b.npz       the seeded head of set B's ValueNet (22 KB, committed)
tools/sweep_vs_reference.py (GPU box) then compares both kernels with these; tools/gen_sweep_golden.py turns
its worst cases into tests/golden/sweep_worst.npz.

    python tools/gen_sweep_reference.py [n_batches=12]
"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv, argv = sys.argv[:1], sys.argv[1:]
sys.path.insert(0, os.path.join(REPO, "tools"))
import gen_golden as G  # noqa: E402  (imports the reference)
import torch  # noqa: E402

from bokego_amd.workload import make_batch  # noqa: E402

N_BATCH = int(argv[0]) if argv else 12
SEED0 = 1_000_000
OUT = os.path.join(REPO, "tests", "golden", "_sweep")
HEAD_KEYS = ("bn.weight", "bn.bias", "bn.running_mean", "bn.running_var", "lin1.weight", "lin1.bias", "lin_bn.weight",
             "lin_bn.bias", "lin_bn.running_mean", "lin_bn.running_var", "lin2.weight", "lin2.bias")


def build_nets_b():
    """set B: the two trained trunks swapped, and a differently seeded value head."""
    W = G.W
    p19 = torch.load(os.path.join(W, "policy_19.pt"), map_location="cpu")["model_state_dict"]
    p17 = torch.load(os.path.join(W, "policy_17.pt"), map_location="cpu")["model_state_dict"]
    pi = G.nnet.PolicyNet()
    pi.load_state_dict(p17)
    pi.eval()
    torch.manual_seed(31337)
    v = G.nnet.ValueNet()
    v.load_policy_dict(p19)
    g = torch.Generator().manual_seed(31338)
    v.bn.running_mean.copy_(torch.randn(1, generator=g) * 0.5)
    v.bn.running_var.copy_(torch.rand(1, generator=g) * 1.5 + 0.5)
    v.bn.weight.copy_(torch.rand(1, generator=g) + 0.5)
    v.bn.bias.copy_(torch.randn(1, generator=g) * 0.3 + 0.5)
    v.lin_bn.running_mean.copy_(torch.randn(64, generator=g) * 0.5)
    v.lin_bn.running_var.copy_(torch.rand(64, generator=g) * 1.5 + 0.5)
    v.lin_bn.weight.copy_(torch.rand(64, generator=g) + 0.5)
    v.lin_bn.bias.copy_(torch.randn(64, generator=g) * 0.3)
    v.lin1.weight.mul_(2.0)
    v.lin2.weight.mul_(2.0)
    v.lin2.bias.add_(0.3)
    v.eval()
    return pi, v


def main():
    os.makedirs(OUT, exist_ok=True)
    sets = {"A": G.build_nets(), "B": build_nets_b()}
    sd = sets["B"][1].state_dict()
    np.savez(os.path.join(REPO, "tests", "golden", "value_head_b.npz"), **{k: sd[k].numpy() for k in HEAD_KEYS})
    acc = {k: {"logits": [], "values": [], "dlogits64": [], "dvalues64": []} for k in sets}
    t0 = time.time()
    for i in range(N_BATCH):
        x8 = make_batch(4096, seed_base=SEED0 + i * 4096, dtype=np.uint8)
        x = torch.from_numpy(x8.astype(np.float32))
        for name, (pi, v) in sets.items():
            lg, va = pi(x).numpy(), v(x).numpy().reshape(-1)
            pi.double(), v.double()
            lg64, va64 = pi(x.double()).numpy(), v(x.double()).numpy().reshape(-1)
            pi.float(), v.float()
            a = acc[name]
            a["logits"].append(lg.astype(np.float32))
            a["values"].append(va.astype(np.float32))
            a["dlogits64"].append((lg64 - lg.astype(np.float64)).astype(np.float32))
            a["dvalues64"].append((va64 - va.astype(np.float64)).astype(np.float32))
        print(f"batch {i + 1}/{N_BATCH}  {time.time() - t0:.0f}s", flush=True)
    for name, a in acc.items():
        np.savez(os.path.join(OUT, f"ref_{name}.npz"), seed0=SEED0, n=N_BATCH * 4096,
                 **{k: np.concatenate(v) for k, v in a.items()})
        d = np.concatenate(a["dlogits64"])
        print(f"set {name}: reference fp32 vs its own float64: max |dlogit| {np.abs(d).max():.3g}, "
              f"max |logit| {np.abs(np.concatenate(a['logits'])).max():.3g}")


if __name__ == "__main__":
    main()
