#!/usr/bin/env python3
"""Same-box bit comparison of two builds of libbokego_amd.so (BK_LIB_PATH selects the library per process):
    python tools/ab_bits.py dump out.npz [precision]    # run the engine on a fixed set of batches, save every output
    python tools/ab_bits.py cmp a.npz b.npz             # exit 1 unless all arrays are bit-identical
Batches cover the 1-, 2- and 3-board workgroup variants, the cooperative small-batch form (8, 4 and 2 CUs per board;
BK_COOP=0 switches it off), split launches and the policy-prefix form.  BK_LIB_ANY_ABI=1 lets an older build load."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "dump":
    from bokego_amd.bkw import load_bkw
    from bokego_amd.engine import LeafEngine
    from bokego_amd.workload import make_batch
    g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    eng = LeafEngine(load_bkw(os.path.join(g, "policy_19.bkw")), load_bkw(os.path.join(g, "value_synth.bkw")), max_batch=4096,
                     precision=sys.argv[3] if len(sys.argv) > 3 else "f32")
    x = make_batch(4096, seed_base=777_000, dtype=np.uint8)
    out = {}
    for B, npol in ((1, 1), (2, 2), (5, 1), (17, 17), (40, 1), (62, 30), (82, 1), (128, 0), (100, 100), (244, 0), (1201, 40), (1500, 1500), (4096, 4096)):
        o = eng.eval(x[:B], logits=npol > 0, probs=npol > 0, value=True, n_policy=npol)
        for k, v in o.items():
            out[f"{k}_{B}_{npol}"] = v
    np.savez(sys.argv[2], **out)
    print("dumped", len(out), "arrays with", os.environ.get("BK_LIB_PATH", "the default library"))
else:
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = [k for k in a.files if not np.array_equal(a[k], b[k])]
    for k in bad:
        print(k, "max |diff|", float(np.abs(a[k] - b[k]).max()))
    print("bit-identical" if not bad else f"{len(bad)} of {len(a.files)} arrays differ")
    sys.exit(1 if bad else 0)
