"""Three boards on 2 / 4 CUs (bk_leaf_eval_coop3_kernel) against the whole-board forms: bits and kernel time per request size.
    python tools/coop3_probe.py [--eight]     (--eight: three boards on EIGHT CUs against the 3- / 2-CUs-per-board forms, 60..96 tasks)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bokego_amd.bkw import load_bkw  # noqa: E402
from bokego_amd.engine import LeafEngine  # noqa: E402
from bokego_amd.workload import make_batch  # noqa: E402

g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
eng = LeafEngine(load_bkw(os.path.join(g, "policy_19.bkw")), load_bkw(os.path.join(g, "value_synth.bkw")), max_batch=1024)
x = make_batch(1024, seed_base=1, dtype=np.uint8)
ref = eng.eval(x, logits=True, probs=True, value=True)          # one big request: 3-board workgroups
eng.set_profiling(True)


def timed(B, npol, reps=20):
    for _ in range(3):
        out = eng.eval(x[:B], logits=True, probs=True, value=True, n_policy=npol)
    s0 = eng.stats()
    for _ in range(reps):
        out = eng.eval(x[:B], logits=True, probs=True, value=True, n_policy=npol)
    s1 = eng.stats()
    n = s1["kernel_ms_count"] - s0["kernel_ms_count"]
    same = (np.array_equal(out["logits"], ref["logits"][:npol]) and np.array_equal(out["probs"], ref["probs"][:npol]) and
            np.array_equal(out["value"], ref["value"][:B]))
    return 1e3 * (s1["kernel_ms_sum"] - s0["kernel_ms_sum"]) / n, s1["coop_launches"] - s0["coop_launches"], s1["coop_fallbacks"] - s0["coop_fallbacks"], same


if "--eight" in sys.argv:
    for B, npol in ((60, 1), (64, 0), (70, 1), (78, 1), (80, 1), (84, 0), (85, 1), (88, 2), (90, 1), (92, 2), (93, 3), (60, 30), (95, 0), (96, 0)):
        row = []
        for forced in (None, "8"):
            eng.set_option("coop3", -1 if forced is None else int(forced))
            us, coop, fb, same = timed(B, npol)
            row.append(f"{'default (CUs per board by tasks)' if forced is None else 'three boards on 8 CUs':>32s} {us:7.1f} us (cooperative launches {coop}, fallbacks {fb}, bits {'equal' if same else 'DIFFER'})")
        print(f"B {B:4d} + {npol:2d} policy rows = {B + npol:4d} tasks: " + " | ".join(row), flush=True)
    sys.exit(0)
for B, npol in ((126, 3), (128, 1), (150, 2), (170, 10), (186, 6), (190, 1), (200, 8), (250, 6), (256, 1), (300, 20), (340, 30), (370, 14), (380, 4), (383, 1), (400, 10), (500, 12)):
    row = []
    for forced in ("0", None):
        eng.set_option("coop3", -1 if forced is None else int(forced))
        us, coop, fb, same = timed(B, npol)
        row.append(f"{'whole-board forms' if forced == '0' else 'default':>17s} {us:7.1f} us (cooperative launches {coop}, fallbacks {fb}, bits {'equal' if same else 'DIFFER'})")
    print(f"B {B:4d} + {npol:2d} policy rows = {B + npol:4d} tasks: " + " | ".join(row), flush=True)
