"""Diagnostic: phases of the cooperative small-batch kernel (stamped build only):
    make -C bokego_amd/csrc diag && BK_LIB_PATH=bokego_amd/libbokego_amd_diag.so python tools/stamp_coop.py [boards] [coop3 form: 2|4|8] [wave]
Shares only (the stamps perturb the kernel)."""
import ctypes, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bokego_amd.bkw import load_bkw
from bokego_amd.engine import LeafEngine
from bokego_amd.workload import make_batch

G = os.path.join(REPO, "tests", "golden")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 62
WAVE = int(sys.argv[3]) if len(sys.argv) > 3 else 0
e = LeafEngine(load_bkw(f"{G}/policy_19.bkw"), load_bkw(f"{G}/value_synth.bkw"), max_batch=512)
if len(sys.argv) > 2:
    e.set_option("coop3", int(sys.argv[2]))          # the three-boards forms (bk_leaf_eval_coop3_kernel)
x = make_batch(B, seed_base=1, dtype=np.uint8)
for _ in range(3):
    e.eval(x, probs=True, value=True, n_policy=1)
assert e.stats()["coop_launches"] == 3
nblk = 1024
buf = np.zeros((nblk, 4, 32), np.uint64)
lib = e._lib
lib.bk_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
assert lib.bk_debug_read_stamps(e._h, buf.ctypes.data, nblk) == 0
t = buf.astype(np.int64)
live = t[:, WAVE, 1] > 0
full = live & (t[:, WAVE, 30] > 0)          # the slices that go on to the heads
print(f"boards {B}: {live.sum()} live workgroups, {full.sum()} with heads")
w = t[full][:, WAVE, :]                  # one wave of those (0: position group 0 / first half)
tot = np.median(w[:, 30] - w[:, 0])
rows = [("staging", w[:, 1] - w[:, 0])]
agg = {"conv": 0, "store + ack": 0, "meet (barrier, arrive, poll, barrier)": 0, "fetch + barrier": 0}
for L in range(7):
    prev = w[:, 1] if L == 0 else w[:, 5 + 4 * (L - 1)]
    agg["conv"] += np.median(w[:, 2 + 4 * L] - prev)
    agg["store + ack"] += np.median(w[:, 3 + 4 * L] - w[:, 2 + 4 * L])
    agg["meet (barrier, arrive, poll, barrier)"] += np.median(w[:, 4 + 4 * L] - w[:, 3 + 4 * L])
    agg["fetch + barrier"] += np.median(w[:, 5 + 4 * L] - w[:, 4 + 4 * L])
print(f"total cycles (median) {tot:.0f}; staging {np.median(rows[0][1]):.0f}; heads {np.median(w[:, 30] - w[:, 29 - 0 if False else 5 + 24]):.0f}")
for k, v in agg.items():
    print(f"  {k:40s} {v:9.0f} cycles  {100 * v / tot:5.1f} %   ({v / 7:7.0f} per layer)")
per_layer_meet = [np.median(w[:, 4 + 4 * L] - w[:, 3 + 4 * L]) for L in range(7)]
print("  meet per layer:", [int(v) for v in per_layer_meet])
print("  conv per layer:", [int(np.median(w[:, 2 + 4 * L] - (w[:, 1] if L == 0 else w[:, 5 + 4 * (L - 1)]))) for L in range(7)])
print("  store per layer:", [int(np.median(w[:, 3 + 4 * L] - w[:, 2 + 4 * L])) for L in range(7)])
print("  fetch per layer:", [int(np.median(w[:, 5 + 4 * L] - w[:, 4 + 4 * L])) for L in range(7)])
pk = w[:, 31]
if (pk > 0).all():
    print("  staging in detail (cycles): stamp(0) -> loads issued = the rest |  zeroing LDS", int(np.median(pk & 0xffff)), "| waiting at the barrier (loads landing)",
          int(np.median((pk >> 16) & 0xffff)), "| scatter", int(np.median((pk >> 32) & 0xffff)), "| all of staging", int(np.median(w[:, 1] - w[:, 0])))
