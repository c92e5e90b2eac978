#!/usr/bin/env python3
"""Where the fp32 kernel's distance from float64 comes from, layer by layer, and what another summation order would buy
(VERDICT r3 item 2).  CPU only: tools/emu/kernel_emu.c reproduces the kernel's summation order term by term (checked bit for
bit against the GPU kernel: tests/test_gpu_parity.py::test_kernel_order_emulation_matches_the_kernel, fixture
tests/golden/emu_check.npz), so variants can be priced on the whole 49,152-position sweep without a GPU.

    python tools/error_budget.py budget           # per-layer budget on the sweep's worst positions + the 536 goldens
    python tools/error_budget.py sweep [A|B] [n]  # variants on the whole sweep (needs tests/golden/_sweep/, ~1 min per variant and set)
    python tools/error_budget.py fixture          # writes tests/golden/emu_check.npz (emulated outputs of 64 positions)
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bokego_amd.bkw import load_bkw  # noqa: E402

G = os.path.join(REPO, "tests", "golden")
EPS = 1e-5


def slot_perm(s):
    return (s & ~15) | (((s >> 1) & 1) << 3) | (((s >> 2) & 1) << 2) | ((s & 1) << 1) | ((s >> 3) & 1)


class EmuNet(ctypes.Structure):
    _fields_ = [("w0", ctypes.c_void_p), ("ch0", ctypes.c_void_p), ("w3", ctypes.c_void_p), ("ch3", ctypes.c_void_p),
                ("bias", ctypes.c_void_p), ("head_w", ctypes.c_void_p), ("head_b", ctypes.c_void_p), ("lin1_wt", ctypes.c_void_p),
                ("lin1_b", ctypes.c_void_p), ("lin2_w", ctypes.c_void_p), ("lin2_b", ctypes.c_float)]


def lib():
    so = os.path.join(REPO, "tools", "emu", "libkemu.so")
    src = os.path.join(REPO, "tools", "emu", "kernel_emu.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O3", "-march=native", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared", src, "-o", so, "-lm"])
    L = ctypes.CDLL(so)
    L.emu_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                            ctypes.c_void_p, ctypes.c_void_p]
    return L


class Net:
    """weights folded exactly as bk_engine.cpp:pack_trunk / load_weights do (float64, rounded once), in chain order"""

    def __init__(self, sd):
        self.keep = []
        f64 = lambda k: np.asarray(sd[k], np.float64)  # noqa: E731
        # the chain's input channels: layer 0 -- group 0: k-steps 0..3 x quads 0..3, group 1: k-steps 0,1 (planes 16..23) and 2
        # (planes 24..26); layers 1..6 -- 8 groups x 4 k-steps x 4 quads
        ch0 = [slot_perm(4 * kq + j) for j in range(4) for kq in range(4)] + [slot_perm(16 + 4 * kq + j) for j in range(2) for kq in range(4)] + [24, 25, 26]
        ch3 = [slot_perm(16 * g + 4 * kq + j) for g in range(8) for j in range(4) for kq in range(4)]
        assert sorted(ch0) == list(range(27)) and sorted(ch3) == list(range(128))
        bias = np.zeros((7, 128), np.float32)
        w3 = np.zeros((6, 9, 128, 128), np.float32)
        for l in range(7):
            cw, cb = f64(f"conv.{3 * l}.weight"), f64(f"conv.{3 * l}.bias")
            bw, bb, bm, bv = (f64(f"conv.{3 * l + 1}.{n}") for n in ("weight", "bias", "running_mean", "running_var"))
            scale = bw / np.sqrt(bv + EPS)
            bias[l] = ((cb - bm) * scale + bb).astype(np.float32)
            wf = (cw * scale[:, None, None, None]).astype(np.float32)          # [co][ci][ky][kx]
            wt = wf.reshape(128, wf.shape[1], -1).transpose(2, 1, 0)            # [tap][ci][co]
            if l == 0:
                w0 = np.ascontiguousarray(wt[:, ch0, :])
            else:
                w3[l - 1] = wt[:, ch3, :]
        hs, hshift = 1.0, 0.0
        if "lin1.weight" in sd:
            s = float(f64("bn.weight")[0] / np.sqrt(f64("bn.running_var")[0] + EPS))
            hs, hshift = s, float(f64("bn.bias")[0] - f64("bn.running_mean")[0] * s)
        head_w = (f64("conv.21.weight").reshape(128) * hs).astype(np.float32)
        head_b = (f64("conv.21.bias").reshape(81) * hs + hshift).astype(np.float32)
        arrs = dict(w0=w0, ch0=np.asarray(ch0, np.int32), w3=w3, ch3=np.asarray(ch3, np.int32), bias=bias, head_w=head_w, head_b=head_b)
        self.c = EmuNet()
        self.c.lin2_b = 0.0
        if "lin1.weight" in sd:
            sj = f64("lin_bn.weight") / np.sqrt(f64("lin_bn.running_var") + EPS)
            arrs["lin1_wt"] = np.ascontiguousarray((f64("lin1.weight") * sj[:, None]).T.astype(np.float32))    # [81][64]
            arrs["lin1_b"] = ((f64("lin1.bias") - f64("lin_bn.running_mean")) * sj + f64("lin_bn.bias")).astype(np.float32)
            arrs["lin2_w"] = np.asarray(sd["lin2.weight"], np.float32).reshape(64).copy()
            self.c.lin2_b = float(np.asarray(sd["lin2.bias"], np.float32).reshape(-1)[0])
        for k, v in arrs.items():
            v = np.ascontiguousarray(v)
            self.keep.append(v)
            setattr(self.c, k, v.ctypes.data)
        self.is_value = "lin1.weight" in sd

    def run(self, planes_u8, two_acc=0, exact=0, head=0, lin=0):
        """-> head outputs [B,81] (policy: logits), and for a value net [B,2] = (pre-tanh, tanh)"""
        x = np.ascontiguousarray(planes_u8, np.uint8).reshape(-1, 2187)
        B = len(x)
        ho, vo = np.empty((B, 81), np.float32), np.empty((B, 2), np.float32)
        lib().emu_batch(ctypes.byref(self.c), x.ctypes.data, B, two_acc, exact, head, lin, ho.ctypes.data, vo.ctypes.data if self.is_value else None)
        return (ho, vo) if self.is_value else ho


def weight_sets():
    p19, vs = load_bkw(os.path.join(G, "policy_19.bkw")), load_bkw(os.path.join(G, "value_synth.bkw"))
    head_b = np.load(os.path.join(G, "value_head_b.npz"))
    pol_b = {k: v for k, v in vs.items() if k.startswith("conv.")}
    val_b = dict(p19)
    val_b.update({k: head_b[k] for k in head_b.files})
    return {"A": (p19, vs), "B": (pol_b, val_b)}


ALL = 0x7F
HALVES = 0x100          # bit 8 of the two-chain mask: split by halves of the window (as the kernel does) instead of by tap parity
VARIANTS = [  # name, two-chain mask, exact mask, head mode, lin mode
    ("rounds 1-3: one chain everywhere", 0, 0, 0, 0),
    ("heads as 4 chains", 0, 0, 1, 1),
    ("heads as 4 chains + layer 6 in two chains (VERDICT r3 item 2c)", HALVES | 1 << 6, 0, 1, 1),
    ("heads as 4 chains + layers 5,6 in two chains", HALVES | 3 << 5, 0, 1, 1),
    ("heads as 4 chains + layers 1..6 in two chains", HALVES | 0x7E, 0, 1, 1),
    ("round 4, shipped: heads as 4 chains + every layer in two chains (window halves)", HALVES | ALL, 0, 1, 1),
    ("... by tap parity instead", ALL, 0, 1, 1),
    ("heads exact (float64), convs one chain", 0, 0, 2, 2),
    ("everything exact (float64 sums of the folded fp32 weights): what folding + fp32 activations alone cost", 0, ALL, 2, 2),
]
SHIPPED = (HALVES | ALL, 0, 1, 1)


def budget():
    sets = weight_sets()
    sw = np.load(os.path.join(G, "sweep_worst.npz"))
    rows = []
    for s in ("A", "B"):
        P, V = Net(sets[s][0]), Net(sets[s][1])
        x = sw[f"features_{s}"]
        lg64, va64, lgref, varef = sw[f"logits_f64_{s}"], sw[f"values_f64_{s}"], sw[f"logits_{s}"], sw[f"values_{s}"]
        def dist(two_acc, exact, head, lin):
            lg = P.run(x, two_acc, exact, head, lin)
            _, vo = V.run(x, two_acc, exact, head, lin)
            return (np.abs(lg - lg64).max(), np.abs(lg - lgref).max(), np.abs(vo[:, 1] - va64).max(), np.abs(vo[:, 1] - varef).max())
        for name, *v in VARIANTS:
            rows.append((s, name, *dist(*v)))
        for l in range(7):
            rows.append((s, f"layer {l} alone exact", *dist(0, 1 << l, 0, 0)))
        for l in range(7):
            rows.append((s, f"all exact BUT layer {l} (one chain)", *dist(0, ALL & ~(1 << l), 2, 2)))
        rows.append((s, "all exact BUT the heads (one chain)", *dist(0, ALL, 0, 0)))
        rows.append((s, "all exact BUT the heads (4 chains)", *dist(0, ALL, 1, 1)))
    print("| set | variant | max dlogit vs float64 | vs reference fp32 | max dvalue vs float64 | vs reference |")
    print("|---|---|---|---|---|---|")
    for r in rows:
        print(f"| {r[0]} | {r[1]} | {r[2]:.3g} | {r[3]:.3g} | {r[4]:.3g} | {r[5]:.3g} |")


def sweep(which, n_max, pick=(0, 1, 2, 5)):
    from bokego_amd.workload import make_batch
    sets = weight_sets()
    for s in which:
        ref = np.load(os.path.join(G, "_sweep", f"ref_{s}.npz"))
        n = min(int(ref["n"]), n_max)
        seed0 = int(ref["seed0"])
        feats = np.concatenate([make_batch(min(4096, n - i), seed_base=seed0 + i, dtype=np.uint8) for i in range(0, n, 4096)])
        lg_ref, va_ref = ref["logits"][:n], ref["values"][:n]
        lg64, va64 = lg_ref.astype(np.float64) + ref["dlogits64"][:n], va_ref.astype(np.float64) + ref["dvalues64"][:n]
        P, V = Net(sets[s][0]), Net(sets[s][1])
        print(f"set {s}: {n} positions, max |logit| {np.abs(lg_ref).max():.1f}; reference fp32 vs float64: {np.abs(ref['dlogits64'][:n]).max():.3g}", flush=True)
        for name, *v in [VARIANTS[i] for i in pick]:
            lg = P.run(feats, *v)
            _, vo = V.run(feats, *v)
            d64, dref = np.abs(lg - lg64).max(1), np.abs(lg - lg_ref).max(1)
            print(f"| {s} | {name} | {d64.max():.3g} (p99.9 {np.quantile(d64, .999):.3g}, mean {d64.mean():.3g}) | {dref.max():.3g} (p99.9 {np.quantile(dref, .999):.3g}) "
                  f"| {np.abs(vo[:, 1] - va64).max():.3g} | {np.abs(vo[:, 1] - va_ref).max():.3g} |", flush=True)


def fixture():
    sets = weight_sets()
    x = np.load(os.path.join(G, "sweep_worst.npz"))["features_A"][:64]
    P, V = Net(sets["A"][0]), Net(sets["A"][1])
    out = {"features": x}
    for tag, v in (("r3", (0, 0, 0, 0)), ("r4", SHIPPED)):
        out[f"logits_{tag}"] = P.run(x, *v)
        out[f"value_pre_tanh_{tag}"] = V.run(x, *v)[1][:, 0]
    np.savez_compressed(os.path.join(G, "emu_check.npz"), **out)
    print("wrote tests/golden/emu_check.npz")


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "budget"
    if cmd == "budget":
        budget()
    elif cmd == "sweep":
        sweep([sys.argv[2]] if len(sys.argv) > 2 and sys.argv[2] in "AB" else ["A", "B"], int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 30)
    else:
        fixture()
