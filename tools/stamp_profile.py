"""Diagnostic: where does a workgroup's time go?  Needs the stamped build:
    make -C bokego_amd/csrc diag && BK_LIB_PATH=bokego_amd/libbokego_amd_diag.so python tools/stamp_profile.py
Reads SHARES only (the stamps serialise the kernel; its run time is not representative)."""
import ctypes, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bokego_amd.bkw import load_bkw
from bokego_amd.engine import LeafEngine
import bench

G = os.path.join(REPO, "tests", "golden")
e = LeafEngine(load_bkw(f"{G}/policy_19.bkw"), load_bkw(f"{G}/value_synth.bkw"), max_batch=4096)
x = bench.make_workload(4096, 1)[0]
for _ in range(3):
    e.eval(x, logits=False, probs=True, value=True)
nblk = 2 * ((4096 + 2) // 3)
buf = np.zeros((nblk, 4, 32), np.uint64)
lib = e._lib
lib.bk_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
assert lib.bk_debug_read_stamps(e._h, buf.ctypes.data, nblk) == 0
t = buf.astype(np.int64)
names = {0: "start", 1: "staged input", 30: "end (heads)"}
for L in range(7):
    names[2 + 4 * L] = f"L{L} conv"; names[3 + 4 * L] = f"L{L} barrier1(+zero)"; names[4 + 4 * L] = f"L{L} store"; names[5 + 4 * L] = f"L{L} barrier2"
order = sorted(names)
tot = (t[:, :, 30] - t[:, :, 0]).astype(np.float64)
print(f"blocks {nblk}; per-wave total cycles: median {np.median(tot):.0f}  p10 {np.percentile(tot,10):.0f}  p90 {np.percentile(tot,90):.0f}")
agg = {}
prev = order[0]
for k in order[1:]:
    d = (t[:, :, k] - t[:, :, prev]).astype(np.float64)
    kind = names[k].split(" ", 1)[1] if names[k].startswith("L") else names[k]
    agg.setdefault(kind, 0.0)
    agg[kind] += np.median(d)
    print(f"  {names[k]:24s} median {np.median(d):10.0f}  ({np.median(d)/np.median(tot)*100:5.2f} %)  p90 {np.percentile(d,90):10.0f}")
    prev = k
print({k: f"{v/np.median(tot)*100:.2f} %" for k, v in agg.items()})
