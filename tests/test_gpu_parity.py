"""-m gpu: the HIP engine (through the C ABI) vs the CPU oracle and the reference's golden vectors."""
import os

import numpy as np
import pytest

from bokego_amd.bkw import load_bkw

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

TOL_LOGIT = 1e-4  # BASELINE.json north_star: "within 1e-4 fp32"
TOL_VALUE = 1e-4
TOL_PROB = 1e-5


@pytest.fixture(scope="module")
def weights():
    return load_bkw(os.path.join(GOLDEN, "policy_19.bkw")), load_bkw(os.path.join(GOLDEN, "value_synth.bkw"))


@pytest.fixture(scope="module", params=["f32", "f16x2"])
def engine(weights, request):
    """every parity test runs on both arithmetic modes of the conv stacks (same tolerances)"""
    from bokego_amd.engine import LeafEngine
    e = LeafEngine(weights[0], weights[1], device_id=0, max_batch=4096, precision=request.param)
    assert e.precision == request.param
    yield e
    e.close()


@pytest.fixture(scope="module")
def oracle(weights):
    from oracle.oracle import OraclePolicy, OracleValue
    return OraclePolicy(weights[0]), OracleValue(weights[1])


@pytest.fixture(scope="module")
def gold():
    f = np.load(os.path.join(GOLDEN, "features.npz"))["incremental"]
    n = np.load(os.path.join(GOLDEN, "nets.npz"))
    return f, n


def _check(out, lg, pr, va):
    assert np.abs(out["logits"] - lg).max() < TOL_LOGIT
    assert np.abs(out["probs"] - pr).max() < TOL_PROB
    assert np.abs(out["value"] - va).max() < TOL_VALUE


def test_config1_single_positions_vs_reference(engine, gold):
    """BASELINE config 1: every golden position one at a time (B=1)."""
    f, n = gold
    for i in range(len(f)):
        out = engine.eval(f[i].astype(np.float32), logits=True, probs=True, value=True)
        _check(out, n["logits_b1"][i:i + 1], n["probs_b1"][i:i + 1], n["values_b1"][i:i + 1])


@pytest.mark.parametrize("B", [1, 2, 3, 4, 5, 7, 64, 243, 244, 536])
def test_batched_vs_reference_goldens(engine, gold, B):
    f, n = gold
    out = engine.eval(f[:B].astype(np.float32), logits=True, probs=True, value=True)
    _check(out, n["logits_b1"][:B], n["probs_b1"][:B], n["values_b1"][:B])


# (positions, the launch form the fp32 planner picks for B policy + B value tasks on 256 CUs): every cooperative form and the
# whole-board forms between them -- 12 / 8 / 6 / 4 / 3 / 2 CUs per board, three boards on 8 / 4 / 2 CUs (108 / 104 / 102), 0 = whole boards
PLANNER_CASES = [(4, 12), (8, 12), (12, 8), (17, 6), (25, 4), (35, 3), (45, 108), (60, 2), (80, 104), (110, 0), (150, 102), (200, 0)]


@pytest.mark.parametrize("B,form", PLANNER_CASES)
def test_every_planner_choice_vs_reference_goldens(engine, gold, B, form):
    """VERDICT r5 weak #1a: the default engine against the REFERENCE's recorded outputs at a request size for every choice the
    launch planner has (the bit-identity tests further down compare the forms with each other, HIP against HIP): the size lands on
    the form intended (bk_plan_query, and the engine's cooperative-launch counter) and the outputs are the reference's within
    north_star's tolerance."""
    from bokego_amd import _lib as L
    f, n = gold
    assert L.load().bk_plan_query(B, B, 256, 0, None) == form
    c0 = engine.stats()["coop_launches"]
    out = engine.eval(f[:B].astype(np.float32), logits=True, probs=True, value=True)
    st = engine.stats()
    if engine.precision == "f32":
        assert st["coop_launches"] - c0 == (1 if form else 0) and st["coop_fallbacks"] == 0
    _check(out, n["logits_b1"][:B], n["probs_b1"][:B], n["values_b1"][:B])


def test_uint8_features_identical(engine, gold):
    f, _ = gold
    a = engine.eval(f.astype(np.float32), logits=True, probs=True, value=True)
    b = engine.eval(f.astype(np.uint8), logits=True, probs=True, value=True)
    for k in ("logits", "probs", "value"):
        assert np.array_equal(a[k], b[k])


def test_playout_positions(engine):
    p = np.load(os.path.join(GOLDEN, "playouts.npz"))
    out = engine.eval(p["features"], logits=True, probs=False, value=True)
    assert np.abs(out["logits"] - p["logits"]).max() < TOL_LOGIT
    assert np.abs(out["value"] - p["values"]).max() < TOL_VALUE


def test_vs_oracle_random_inputs(engine, oracle, weights):
    """iid random planes (values 0..7): far from the golden distribution.  Logits on such inputs are much larger than on real
    positions (|logit| up to ~110), so the bound scales with them.  Ground truth is the float64 evaluation of the reference's
    operators (oracle/torch_ref.py in double): against it the kernel must stay within the scaled tolerance -- two fp32
    evaluations that sum in different orders (the kernel's two chains per dot product, the C oracle's single one) each carry
    their own rounding error, so between THEM the bound is 1.5x for the logits; for the value -- both within north_star's 1e-4 of
    the float64 evaluation -- it is the sum of the two, 2e-4 (until round 5 this line said 1e-3 without saying why)."""
    import torch
    from oracle.torch_ref import TorchPolicy, TorchValue
    rng = np.random.default_rng(7)
    x = rng.integers(0, 8, size=(96, 27, 9, 9)).astype(np.float32)
    out = engine.eval(x, logits=True, probs=False, value=True)
    xd = torch.from_numpy(x).double()
    lg64 = TorchPolicy(weights[0]).double()(xd).numpy()
    va64 = TorchValue(weights[1]).double()(xd).numpy()
    scale = max(1.0, np.abs(lg64).max() / 50.0)
    assert np.abs(out["logits"] - lg64).max() < TOL_LOGIT * scale
    assert np.abs(out["value"] - va64).max() < TOL_VALUE
    assert np.abs(out["logits"] - oracle[0](x)).max() < 1.5 * TOL_LOGIT * scale
    assert np.abs(oracle[1](x) - va64).max() < TOL_VALUE                 # (the oracle's own distance from float64)
    assert np.abs(out["value"] - oracle[1](x)).max() < 2 * TOL_VALUE


def test_full_batch_4096_properties(engine, gold, oracle):
    """BASELINE config 2 size: batch invariance + slot invariance + oracle spot check."""
    f, n = gold
    rng = np.random.default_rng(11)
    idx = rng.integers(0, len(f), size=4096)
    x = f[idx].astype(np.float32)
    out = engine.eval(x, logits=True, probs=True, value=True)
    # every copy of a position gives the same answer as its B=1 golden, wherever it sits in the batch
    _check(out, n["logits_b1"][idx], n["probs_b1"][idx], n["values_b1"][idx])
    # bit-exact slot invariance: same input -> same bits regardless of its slot in the batch
    first = {}
    for j, i in enumerate(idx):
        if i in first:
            assert np.array_equal(out["logits"][j], out["logits"][first[i]])
            assert out["value"][j] == out["value"][first[i]]
        else:
            first[i] = j
    assert np.allclose(out["probs"].sum(1), 1.0, atol=1e-5)
    sel = rng.integers(0, 4096, size=64)
    assert np.abs(out["logits"][sel] - oracle[0](x[sel])).max() < TOL_LOGIT


def test_single_net_engines(weights, gold):
    from bokego_amd.engine import LeafEngine
    f, n = gold
    pe = LeafEngine(policy_sd=weights[0], max_batch=64)
    ve = LeafEngine(value_sd=weights[1], max_batch=64)
    x = f[:50].astype(np.float32)
    o = pe.eval(x, logits=True, probs=True, value=False)
    assert np.abs(o["logits"] - n["logits_b1"][:50]).max() < TOL_LOGIT
    v = ve.eval(x, logits=False, probs=False, value=True)
    assert np.abs(v["value"] - n["values_b1"][:50]).max() < TOL_VALUE
    with pytest.raises(ValueError):
        pe.eval(x, logits=False, probs=False, value=True)      # BK_ERR_NO_NET
    with pytest.raises(ValueError):
        pe.eval(f[:65].astype(np.float32), probs=True, value=False)  # BK_ERR_BATCH
    pe.close()
    ve.close()


def test_async_tickets(engine, gold):
    f, n = gold
    x = f.astype(np.float32)
    t = [engine.submit(x[i * 100:(i + 1) * 100], logits=True, probs=False, value=True) for i in range(4)]
    for i in reversed(range(4)):
        o = engine.wait(t[i])
        assert np.abs(o["logits"] - n["logits_b1"][i * 100:(i + 1) * 100]).max() < TOL_LOGIT
        assert np.abs(o["value"] - n["values_b1"][i * 100:(i + 1) * 100]).max() < TOL_VALUE


def test_device_resident_path(engine, gold):
    import torch
    f, n = gold
    x = torch.from_numpy(f[:300].astype(np.float32)).cuda()
    o = engine.eval_device(x, logits=True, probs=True, value=True)
    torch.cuda.synchronize()
    _check({k: o[k].cpu().numpy() for k in ("logits", "probs", "value")},
           n["logits_b1"][:300], n["probs_b1"][:300], n["values_b1"][:300])


def test_f16x2_overflow_falls_back_to_fp32(weights, oracle):
    """Activations beyond the fp16 range: the f16x2 kernel raises its flag and the engine redoes the
    request on the exact fp32 kernel, so the caller still gets fp32-quality results."""
    from bokego_amd.engine import LeafEngine
    rng = np.random.default_rng(3)
    x = (rng.integers(0, 8, size=(5, 27, 9, 9)) * 4000.0).astype(np.float32)   # absurdly large planes
    fast = LeafEngine(weights[0], weights[1], max_batch=8, precision="f16x2")
    exact = LeafEngine(weights[0], weights[1], max_batch=8, precision="f32")
    a = fast.eval(x, logits=True, probs=False, value=True)
    b = exact.eval(x, logits=True, probs=False, value=True)
    assert fast.stats()["f16_overflow_fallbacks"] == 1 and exact.stats()["f16_overflow_fallbacks"] == 0
    assert np.array_equal(a["logits"], b["logits"]) and np.array_equal(a["value"], b["value"])
    lg = oracle[0](x)
    assert np.abs(a["logits"] - lg).max() <= 2e-6 * np.abs(lg).max()
    # ordinary inputs do not trigger it
    fast.eval((x / 4000.0).astype(np.float32), logits=True, probs=False, value=True)
    assert fast.stats()["f16_overflow_fallbacks"] == 1
    fast.close(); exact.close()


def test_f16x2_overflow_on_the_device_path_is_redone_in_fp32(weights, gold):
    """bk_eval_device* (LeafEngine.eval_device, HipPolicyNet/HipValueNet on CUDA tensors): an f16x2 call whose
    kernel overflows is redone by the gated fp32 launch enqueued behind it on the caller's stream -- the
    caller gets bit-identical fp32 results, never the clamped ones, with no host round trip; the calls
    before and after it stay on the f16x2 kernel."""
    import torch
    from bokego_amd.engine import LeafEngine
    from bokego_amd.nnet import HipPolicyNet, HipValueNet
    rng = np.random.default_rng(3)
    big = (rng.integers(0, 8, size=(700, 27, 9, 9)) * 4000.0).astype(np.float32)
    sane = gold[0][:700].astype(np.float32)
    sane[:5] = big[:5] / 4000.0
    fast = LeafEngine(weights[0], weights[1], max_batch=1024, precision="f16x2")
    exact = LeafEngine(weights[0], weights[1], max_batch=1024, precision="f32")
    side = torch.cuda.Stream()
    for stream in (None, side):          # torch's current stream, and an explicit non-default one
        with torch.cuda.stream(side) if stream is not None else torch.cuda.stream(torch.cuda.current_stream()):
            n0 = fast.stats()["f16_device_overflow"]
            xs, xb = torch.from_numpy(sane).cuda(), torch.from_numpy(big).cuda()
            a0 = fast.eval_device(xs, logits=True, probs=True, value=True)
            a1 = fast.eval_device(xb, logits=True, probs=True, value=True, n_policy=333)   # split-launch sized
            a2 = fast.eval_device(xs, logits=True, probs=True, value=True)
            b1 = exact.eval_device(xb, logits=True, probs=True, value=True, n_policy=333)
            f0 = fast.eval(sane, logits=True, probs=True, value=True)
            torch.cuda.synchronize()
            for k in ("logits", "probs", "value"):
                assert torch.equal(a1[k], b1[k]), k                      # redone: the fp32 kernel's bits
                assert torch.equal(a0[k], a2[k]), k                      # neighbours untouched ...
                assert np.array_equal(a0[k].cpu().numpy(), f0[k]), k     # ... and still the f16x2 kernel's bits
            assert torch.isfinite(a1["logits"]).all()
            assert fast.stats()["f16_device_overflow"] == n0 + 1 and exact.stats()["f16_device_overflow"] == 0
    # the drop-in nets on CUDA tensors go through the same path
    pn, vn = HipPolicyNet(weights[0], max_batch=64, precision="f16x2"), HipValueNet(weights[1], max_batch=64, precision="f16x2")
    assert pn.engine().precision == "f16x2" and HipPolicyNet(weights[0]).engine().precision == "f32"   # f32 is the default
    xb = torch.from_numpy(big[:5]).cuda()
    lg, va = pn(xb), vn(xb)
    torch.cuda.synchronize()
    ref = exact.eval(big[:5], logits=True, probs=False, value=True)
    assert lg.is_cuda and np.array_equal(lg.cpu().numpy(), ref["logits"])
    assert np.array_equal(va.cpu().numpy().reshape(-1), ref["value"])
    assert pn.engine().stats()["f16_device_overflow"] == 1 and vn.engine().stats()["f16_device_overflow"] == 1
    fast.close(); exact.close()


def test_abi_edge_cases(weights, gold):
    """Empty batch, max batch, argument errors through the raw C ABI."""
    import ctypes
    from bokego_amd import _lib
    from bokego_amd.engine import LeafEngine
    f, n = gold
    e = LeafEngine(weights[0], weights[1], max_batch=7)
    lib, h = e._lib, e._h
    empty = np.zeros((0, 27, 9, 9), np.float32)
    o = e.eval(empty, logits=True, probs=True, value=True)             # B = 0 is a no-op
    assert o["logits"].shape == (0, 81) and o["value"].shape == (0,)
    x = f[:7].astype(np.float32)
    o = e.eval(x, logits=True, probs=True, value=True)                 # B == max_batch
    assert np.abs(o["logits"] - n["logits_b1"][:7]).max() < TOL_LOGIT
    fp = ctypes.c_void_p
    out = np.empty((7, 81), np.float32)
    assert lib.bk_eval(h, x.ctypes.data, 7, 0, None, None, None) == -1                 # empty want mask
    assert lib.bk_eval(h, x.ctypes.data, 7, _lib.BK_WANT_PROBS, None, None, None) == -1  # missing output buffer
    assert lib.bk_eval(h, None, 7, _lib.BK_WANT_PROBS, None, out.ctypes.data, None) == -1  # missing input
    assert lib.bk_eval(h, x.ctypes.data, 8, _lib.BK_WANT_PROBS, None, out.ctypes.data, None) == -4  # BK_ERR_BATCH
    assert b"max_batch" in lib.bk_last_error(h)
    assert lib.bk_wait(h, 12345) == -1                                                  # unknown ticket
    assert lib.bk_engine_set_precision(h, 7) == -1
    # more tickets than BK_MAX_INFLIGHT
    tickets = [e.submit(x, probs=True, value=True) for _ in range(_lib.BK_MAX_INFLIGHT)]
    with pytest.raises(ValueError):
        e.submit(x, probs=True, value=True)
    for t in tickets:
        e.wait(t)
    s = e.stats()
    assert s["evals"] >= 7 * (1 + _lib.BK_MAX_INFLIGHT) and s["max_batch_seen"] == 7
    e.close()


def test_non_integer_inputs_use_the_lo_half(weights, oracle):
    """Planes that are not exactly representable in fp16 (the reference never produces them, a generic
    caller might): layer 0 must then keep the x_lo product; checked against the oracle."""
    from bokego_amd.engine import LeafEngine
    rng = np.random.default_rng(5)
    x = (rng.integers(0, 8, size=(7, 27, 9, 9)) * (rng.random((7, 27, 9, 9)) < 0.12)).astype(np.float32)
    x += (rng.random(x.shape).astype(np.float32) * 1e-3) * (x > 0)       # 7.0003..., needs 22 bits
    e = LeafEngine(weights[0], weights[1], max_batch=8, precision="f16x2")
    out = e.eval(x, logits=True, probs=False, value=True)
    lg = oracle[0](x)
    assert np.abs(out["logits"] - lg).max() < 1e-4 * max(1.0, np.abs(lg).max() / 50)
    assert np.abs(out["value"] - oracle[1](x)).max() < 1e-4
    e.close()


def test_c_abi_from_plain_c(gold):
    """examples/bk_demo.c: a C program (no Python, no torch) loads BKW1 weights, encodes the empty board with
    libbkgo, evaluates it through bk_eval and prints the SURVEY 8c known answers."""
    import subprocess
    from conftest import REPO
    exe = os.path.join(REPO, "examples", "bk_demo")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "bokego_amd", "csrc"), "demo"])
    out = subprocess.run([exe, os.path.join(GOLDEN, "policy_19.bkw"), os.path.join(GOLDEN, "value_synth.bkw")],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "best move E5 (index 40)" in out.stdout
    p = float(out.stdout.split("p=")[1].split()[0])
    v = float(out.stdout.split("value=")[1].split()[0])
    f, n = gold
    assert abs(p - 0.818645) < 1e-5 and abs(v - float(n["values_b1"][0])) < 1e-4


def test_hard_positions_worst_of_49k(engine):
    """The 48 positions, out of 49,152 seeded random playouts, on which the f16x2 and the exact-fp32 kernel
    differ most (tools/precision_sweep.py), with the reference's own outputs (tools/gen_hard_golden.py);
    |logit| reaches 65 there.  Both kernels must stay inside the tolerance against the reference, and no
    further from the float64 ground truth than 2x the reference's own fp32 rounding."""
    h = np.load(os.path.join(GOLDEN, "hard_positions.npz"))
    out = engine.eval(h["features"], logits=True, probs=False, value=True)
    assert np.abs(out["logits"] - h["logits"]).max() < TOL_LOGIT
    assert np.abs(out["value"] - h["values"]).max() < TOL_VALUE
    ref_noise = np.abs(h["logits"] - h["logits_f64"]).max()
    assert np.abs(out["logits"] - h["logits_f64"]).max() < 2 * ref_noise


def _random_nets(seed, gain):
    """Seeded random PolicyNet / ValueNet tensors with the reference's names and shapes (nnet.py:31-57,73-101):
    conv weights ~ N(0, gain^2 / fan_in), non-trivial BatchNorm statistics everywhere."""
    rng = np.random.default_rng(seed)

    def trunk():
        t = {}
        for l, (c, b) in enumerate(zip((0, 3, 6, 9, 12, 15, 18), (1, 4, 7, 10, 13, 16, 19))):
            cin, k = (27, 5) if l == 0 else (128, 3)
            t[f"conv.{c}.weight"] = (rng.standard_normal((128, cin, k, k)) * gain / np.sqrt(cin * k * k)).astype(np.float32)
            t[f"conv.{c}.bias"] = (rng.standard_normal(128) * 0.1).astype(np.float32)
            t[f"conv.{b}.weight"] = rng.uniform(0.5, 1.5, 128).astype(np.float32)
            t[f"conv.{b}.bias"] = (rng.standard_normal(128) * 0.2).astype(np.float32)
            t[f"conv.{b}.running_mean"] = (rng.standard_normal(128) * 0.2).astype(np.float32)
            t[f"conv.{b}.running_var"] = rng.uniform(0.5, 2.0, 128).astype(np.float32)
        t["conv.21.weight"] = (rng.standard_normal((1, 128, 1, 1)) / np.sqrt(128)).astype(np.float32)
        t["conv.21.bias"] = (rng.standard_normal((1, 9, 9)) * 0.1).astype(np.float32)
        return t

    p, v = trunk(), trunk()
    v.update({"bn.weight": np.float32([1.3]), "bn.bias": np.float32([0.2]), "bn.running_mean": np.float32([-0.1]),
              "bn.running_var": np.float32([0.7]),
              "lin1.weight": (rng.standard_normal((64, 81)) / 9).astype(np.float32), "lin1.bias": (rng.standard_normal(64) * 0.1).astype(np.float32),
              "lin_bn.weight": rng.uniform(0.5, 1.5, 64).astype(np.float32), "lin_bn.bias": (rng.standard_normal(64) * 0.1).astype(np.float32),
              "lin_bn.running_mean": (rng.standard_normal(64) * 0.2).astype(np.float32), "lin_bn.running_var": rng.uniform(0.5, 2.0, 64).astype(np.float32),
              "lin2.weight": (rng.standard_normal((1, 64)) / 8).astype(np.float32), "lin2.bias": np.float32([0.05])})
    return p, v


@pytest.mark.parametrize("precision", ["f32", "f16x2"])
@pytest.mark.parametrize("seed,gain", [(1, 1.0), (2, 1.6), (3, 0.5)])
def test_random_weights_vs_oracle(gold, precision, seed, gain):
    """Nets the kernels were never tuned on (other weight scales -> other per-layer fp16 scale exponents, other
    BatchNorm folds): outputs still agree with the CPU oracle at fp32 level, relative to the logit magnitude."""
    from bokego_amd.engine import LeafEngine
    from oracle.oracle import OraclePolicy, OracleValue
    pw, vw = _random_nets(seed, gain)
    x = gold[0][::7].astype(np.float32)
    eng = LeafEngine(pw, vw, max_batch=128, precision=precision)
    out = eng.eval(x, logits=True, probs=True, value=True)
    st = eng.stats()
    eng.close()
    lg, pr = OraclePolicy(pw)(x, want_probs=True)
    va = OracleValue(vw)(x)
    scale = max(1.0, float(np.abs(lg).max()) / 15.0)      # the goldens' logits reach ~15 at TOL_LOGIT
    assert np.abs(out["logits"] - lg).max() < TOL_LOGIT * scale
    assert np.abs(out["probs"] - pr).max() < TOL_PROB * 2
    assert np.abs(out["value"] - va).max() < TOL_VALUE
    assert np.isfinite(out["logits"]).all() and st["f16_overflow_fallbacks"] in (0, 1)


@pytest.mark.parametrize("precision", ["f16x2", "f32"])
def test_split_launch_is_bit_identical(precision):
    """Mid-size batches run as whole rounds of 3-board workgroups + a tail launch with smaller workgroups
    (bk_stats().split_launches); every output must be bit-identical to the single-launch result."""
    from bokego_amd.engine import LeafEngine
    from bokego_amd.workload import make_batch
    x = make_batch(1500, seed_base=90_000, dtype=np.uint8)
    pw, vw = load_bkw(os.path.join(GOLDEN, "policy_19.bkw")), load_bkw(os.path.join(GOLDEN, "value_synth.bkw"))
    eng = LeafEngine(pw, vw, max_batch=2048, precision=precision)
    outs = {}
    for split in (True, False):
        if not split:
            eng.set_option("no_split", 1)
        res = []
        for B, npol in ((1500, 1500), (1500, 40), (1201, 0), (900, 900), (771, 3), (1000, 0)):
            s0 = eng.stats()["split_launches"]
            res.append(eng.eval(x[:B], logits=npol > 0, probs=npol > 0, value=True, n_policy=npol))
            res[-1]["split"] = eng.stats()["split_launches"] - s0
        outs[split] = res
    assert sum(r["split"] for r in outs[True]) >= 3 and sum(r["split"] for r in outs[False]) == 0
    for a, b in zip(outs[True], outs[False]):
        for k in ("logits", "probs", "value"):
            if k in b:
                assert np.array_equal(a[k], b[k]), k


def _sweep_weight_sets():
    """set A = the goldens' nets; set B = the two trained trunks swapped + another seeded value head
    (tools/gen_sweep_reference.py; value_synth.bkw carries every policy_17 tensor)."""
    p19, vs = load_bkw(os.path.join(GOLDEN, "policy_19.bkw")), load_bkw(os.path.join(GOLDEN, "value_synth.bkw"))
    head_b = np.load(os.path.join(GOLDEN, "value_head_b.npz"))
    val_b = dict(p19)
    val_b.update({k: head_b[k] for k in head_b.files})
    return {"A": (p19, vs), "B": ({k: v for k, v in vs.items() if k.startswith("conv.")}, val_b)}


@pytest.mark.parametrize("precision", ["f32", "f16x2"])
@pytest.mark.parametrize("wset", ["A", "B"])
def test_sweep_worst_cases_vs_reference(precision, wset):
    """The worst positions of the 49,152-position x 2-weight-set sweep AGAINST THE REFERENCE (not kernel vs kernel):
    tools/gen_sweep_reference.py ran the reference (fp32 and float64) over the whole sweep, tools/sweep_vs_reference.py
    both kernels on the GPU, and the positions with the largest |dlogit| / |dvalue| / |dprob| of either kernel became
    this fixture.  tests/golden/sweep_summary.json records the sweep-wide worst case: every figure is < 1e-4."""
    import json
    from bokego_amd.engine import LeafEngine
    w = np.load(os.path.join(GOLDEN, "sweep_worst.npz"))
    pw, vw = _sweep_weight_sets()[wset]
    eng = LeafEngine(pw, vw, max_batch=256, precision=precision)
    out = eng.eval(w[f"features_{wset}"], logits=True, probs=True, value=True)
    st = eng.stats()
    eng.close()
    lg, va = w[f"logits_{wset}"], w[f"values_{wset}"]
    dl = np.abs(out["logits"] - lg).max()
    assert dl < TOL_LOGIT and np.abs(out["value"] - va).max() < TOL_VALUE
    e = np.exp(lg.astype(np.float64) - lg.max(1, keepdims=True))
    assert np.abs(out["probs"] - e / e.sum(1, keepdims=True)).max() < TOL_PROB
    assert st["f16_overflow_fallbacks"] == 0
    # what the sweep recorded for this kernel on the WHOLE sweep is reproduced here on its worst positions
    rec = json.load(open(os.path.join(GOLDEN, "sweep_summary.json")))[wset][precision]
    assert rec["dlogit_vs_reference"]["positions_over_1e-4"] == 0 and rec["dvalue_vs_reference"]["positions_over_1e-4"] == 0
    assert abs(dl - rec["dlogit_vs_reference"]["max"]) < 2e-5
    # no further from the float64 ground truth than 2x the reference's own fp32 rounding
    ref_noise = np.abs(lg - w[f"logits_f64_{wset}"]).max()
    assert np.abs(out["logits"] - w[f"logits_f64_{wset}"]).max() < 2 * ref_noise + 2e-5


@pytest.mark.parametrize("precision", ["f32", "f16x2"])
def test_every_workgroup_size_gives_the_same_bits(precision):
    """1-, 2- and 3-board workgroups (option force_nb) are three different code paths -- different tile sets, edge-class tap
    skipping for 2 and 3 boards, different wave grids -- and must agree bit for bit on every output, for ragged batch
    sizes and policy prefixes."""
    from bokego_amd.engine import LeafEngine
    from bokego_amd.workload import make_batch
    x = make_batch(700, seed_base=123_000, dtype=np.uint8)
    pw, vw = load_bkw(os.path.join(GOLDEN, "policy_19.bkw")), load_bkw(os.path.join(GOLDEN, "value_synth.bkw"))
    eng = LeafEngine(pw, vw, max_batch=1024, precision=precision)
    eng.set_option("no_split", 1)
    outs = {}
    for nb in (1, 2, 3):
        eng.set_option("force_nb", nb)
        outs[nb] = [eng.eval(x[:B], logits=npol > 0, probs=npol > 0, value=True, n_policy=npol)
                    for B, npol in ((1, 1), (2, 2), (5, 3), (82, 1), (163, 163), (700, 700), (697, 50))]
    eng.close()
    for nb in (2, 3):
        for a, b in zip(outs[1], outs[nb]):
            for k in b:
                assert np.array_equal(a[k], b[k]), (nb, k)


def _small_batches(eng, x):
    return [eng.eval(x[:B], logits=npol > 0, probs=npol > 0, value=True, n_policy=npol)
            for B, npol in ((1, 1), (1, 0), (2, 2), (5, 3), (9, 9), (16, 1), (31, 0), (40, 1), (62, 1), (63, 30), (64, 64), (100, 28), (128, 0))]


def test_cooperative_small_batch_form_gives_the_same_bits():
    """Small fp32 batches on the engine's stream run with 12 / 8 / 6 / 4 / 3 / 2 CUs per board (cooperative kernel: output
    channels split 8, 4 or 2 ways and / or the board's points split 3 ways;
    bk_stats().coop_launches): every output bit-identical to the one-CU-per-board form (option coop = 0), for each slice
    count forced over the whole range as well as for the engine's own choice, with no fallback taken."""
    from bokego_amd.engine import LeafEngine
    from bokego_amd.workload import make_batch
    x = make_batch(128, seed_base=321_000, dtype=np.uint8)
    pw, vw = load_bkw(os.path.join(GOLDEN, "policy_19.bkw")), load_bkw(os.path.join(GOLDEN, "value_synth.bkw"))
    eng = LeafEngine(pw, vw, max_batch=256)
    eng.set_option("coop", 0)
    ref = _small_batches(eng, x)
    assert eng.stats()["coop_launches"] == 0
    for mode in (None, "2", "3", "4", "6", "8", "12"):
        if mode is None:
            eng.set_option("coop", -1)
        else:
            eng.set_option("coop", int(mode))
        c0 = eng.stats()["coop_launches"]
        got = _small_batches(eng, x)
        assert eng.stats()["coop_launches"] - c0 >= (13 if mode in (None, "2", "3") else 6), mode
        for a, b in zip(ref, got):
            for k in a:
                assert np.array_equal(a[k], b[k]), (mode, k, a[k].shape)
    assert eng.stats()["coop_fallbacks"] == 0
    # f32 feature planes through the same path; and the goldens' empty board
    xf = x[:40].astype(np.float32)
    eng.set_option("coop", -1)
    a = eng.eval(xf, logits=True, probs=True, value=True)
    eng.set_option("coop", 0)
    b = eng.eval(xf, logits=True, probs=True, value=True)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    eng.close()


def test_cooperative_form_repeats_exactly_under_load():
    """The meeting points of the cooperative kernel hold under repetition and with a second engine's large launches
    competing for the CUs from another stream: 300 small evaluations of varying size, every one equal to the first
    answer for its size."""
    from bokego_amd.engine import LeafEngine
    from bokego_amd.workload import make_batch
    x = make_batch(2048, seed_base=99_000, dtype=np.uint8)
    pw, vw = load_bkw(os.path.join(GOLDEN, "policy_19.bkw")), load_bkw(os.path.join(GOLDEN, "value_synth.bkw"))
    eng, other = LeafEngine(pw, vw, max_batch=256), LeafEngine(pw, vw, max_batch=2048)
    sizes = (1, 7, 33, 62, 70, 90)
    first = {B: eng.eval(x[:B], probs=True, value=True, n_policy=min(B, 3)) for B in sizes}
    pending = []
    for i in range(300):
        if i % 10 == 0:                       # a 2048-board launch of the other engine in flight (its own stream)
            pending.append(other.submit(x, probs=True, value=True))
        B = sizes[i % len(sizes)]
        got = eng.eval(x[:B], probs=True, value=True, n_policy=min(B, 3))
        for k in got:
            assert np.array_equal(got[k], first[B][k]), (i, B, k)
        if len(pending) > 2:
            other.wait(pending.pop(0))
    for t in pending:
        other.wait(t)
    st = eng.stats()
    assert st["coop_launches"] >= 300
    eng.close()
    other.close()


def test_three_boards_on_two_or_four_cus_give_the_same_bits():
    """Requests between the whole-board forms' ranges (129..192 and 257..384 tasks) run as groups of three boards of one net
    shared by 4 resp. 2 CUs (bk_leaf_eval_coop3_kernel: output channels split, the 3-board tile set, the cooperative
    exchange): every output bit-identical to the whole-board forms (BK_COOP3=0), partial groups, policy rows and both forced
    forms included.  (A deserting peer ending in the usual fallback: tests/test_gpu_hooks.py, on the fault-injection build.)"""
    from bokego_amd.engine import LeafEngine
    from bokego_amd.workload import make_batch
    x = make_batch(384, seed_base=77_000, dtype=np.uint8)
    pw, vw = load_bkw(os.path.join(GOLDEN, "policy_19.bkw")), load_bkw(os.path.join(GOLDEN, "value_synth.bkw"))
    eng = LeafEngine(pw, vw, max_batch=512)
    shapes = ((128, 1), (131, 0), (150, 2), (170, 10), (186, 6), (256, 1), (257, 0), (299, 31), (340, 30), (378, 6))
    eng.set_option("coop3", 0)
    ref = [eng.eval(x[:B], logits=True, probs=True, value=True, n_policy=npol) for B, npol in shapes]
    assert eng.stats()["coop_launches"] == 0
    for mode in (None, "2", "4"):
        if mode is None:
            eng.set_option("coop3", -1)
        else:
            eng.set_option("coop3", int(mode))
        c0 = eng.stats()["coop_launches"]
        for (B, npol), want in zip(shapes, ref):
            if mode == "4" and B + npol > 192:
                continue
            got = eng.eval(x[:B], logits=True, probs=True, value=True, n_policy=npol)
            for k in want:
                assert np.array_equal(want[k], got[k]), (mode, B, npol, k)
        assert eng.stats()["coop_launches"] - c0 == (10 if mode != "4" else 5), mode
    assert eng.stats()["coop_fallbacks"] == 0
    # round 5: three boards on EIGHT CUs (one 16-cout tile of all 243 points per workgroup, four waves of four tiles): up to 32
    # groups, i.e. 96 tasks; forced here over its whole range and below it -- the same bits as the whole-board forms
    small = ((1, 1), (5, 2), (31, 0), (60, 30), (80, 1), (84, 0), (88, 2), (93, 3), (96, 0))
    eng.set_option("coop", 0)
    ref8 = [eng.eval(x[:B], logits=True, probs=True, value=True, n_policy=npol) for B, npol in small]
    eng.set_option("coop", -1)
    eng.set_option("coop3", 8)
    c0 = eng.stats()["coop_launches"]
    for (B, npol), want in zip(small, ref8):
        got = eng.eval(x[:B], logits=True, probs=True, value=True, n_policy=npol)
        for k in want:
            assert np.array_equal(want[k], got[k]), ("8", B, npol, k)
    assert eng.stats()["coop_launches"] - c0 == len(small) and eng.stats()["coop_fallbacks"] == 0
    got = eng.eval(x[:95], logits=True, probs=True, value=True, n_policy=1)       # 1 + 32 groups: does not fit, the usual forms run
    with eng.options(coop=0):
        plain = eng.eval(x[:95], logits=True, probs=True, value=True, n_policy=1)
    assert all(np.array_equal(got[k], plain[k]) for k in got)
    eng.set_option("coop3", -1)
    eng.close()


def test_f16x2_overflow_flags_are_per_call_across_caller_streams(weights, gold):
    """two overflowing bk_eval_device calls on two caller streams, enqueued back to back so that they can run
    concurrently (the later call's kernel may raise its flag first).  Every call owns a flag word, so both are redone in
    fp32; with one shared word the earlier call's atomicMax was a no-op and its clamped outputs were kept."""
    import torch
    from bokego_amd.engine import LeafEngine
    rng = np.random.default_rng(11)
    bigs = [(rng.integers(0, 8, size=(n, 27, 9, 9)) * 4000.0).astype(np.float32) for n in (1500, 90, 700, 333)]
    fast = LeafEngine(weights[0], weights[1], max_batch=2048, precision="f16x2")
    exact = LeafEngine(weights[0], weights[1], max_batch=2048, precision="f32")
    streams = [torch.cuda.Stream() for _ in bigs]
    xs = [torch.from_numpy(b).cuda() for b in bigs]
    torch.cuda.synchronize()
    outs = []
    for x, st in zip(xs, streams):                      # no synchronisation between the calls
        with torch.cuda.stream(st):
            outs.append(fast.eval_device(x, logits=True, probs=True, value=True))
    torch.cuda.synchronize()
    for x, o in zip(xs, outs):
        ref = exact.eval_device(x, logits=True, probs=True, value=True)
        torch.cuda.synchronize()
        for k in ("logits", "probs", "value"):
            assert torch.equal(o[k], ref[k]), k
    assert fast.stats()["f16_device_overflow"] == len(bigs)
    fast.close()
    exact.close()


@pytest.mark.parametrize("precision", ["f32", "f16x2"])
def test_large_host_requests_launched_in_two_parts_give_the_same_bits(weights, precision):
    """A large request of host planes is staged by a pool of copy threads and launched in two parts (the first 768
    positions run while the rest is still on its way: bk_engine.cpp submit_common).  Outputs are bit-identical to the
    single-launch path (options no_head_part = 1, copy_threads = 0) for f32 and u8 planes, whole batches and policy prefixes on
    either side of the split, pipelined tickets included."""
    from bokego_amd.engine import LeafEngine
    from bokego_amd.workload import make_batch
    x8 = make_batch(4096, seed_base=77_000, dtype=np.uint8)
    eng = LeafEngine(weights[0], weights[1], max_batch=4096, precision=precision)
    cases = [(4096, 4096, np.float32), (4096, 100, np.float32), (2305, 1000, np.float32), (2304, 768, np.float32), (4000, 0, np.float32)]
    cases.append((4096, 4096, np.uint8))        # 9 MB: below the two-part threshold, the pool still stages it
    def run(B, npol, dt):
        x = x8[:B].astype(dt)
        return eng.eval(x, logits=npol > 0, probs=npol > 0, value=True, n_policy=npol)
    eng.set_option("no_head_part", 1)
    eng.set_option("copy_threads", 0)
    ref = {c: run(*c) for c in cases}
    eng.set_option("no_head_part", 0)
    eng.set_option("copy_threads", 6)
    for c in cases:
        got = run(*c)
        for k in ref[c]:
            assert np.array_equal(ref[c][k], got[k]), (c, k)
    # three tickets in flight, each launched in two parts
    xf = x8.astype(np.float32)
    ts = [eng.submit(xf, logits=True, probs=True, value=True) for _ in range(3)]
    for t in ts:
        got = eng.wait(t)
        for k in ref[cases[0]]:
            assert np.array_equal(ref[cases[0]][k], got[k]), k
    assert eng.stats()["evals"] >= 3 * 4096
    eng.close()


def test_small_requests_without_copies_give_the_same_bits(weights):
    """Small fp32 requests skip both copies ("direct": the encoder reads position records from the pinned slot, the leaf
    kernel writes flag + outputs into the pinned output block).  Same bits as with the copies (option no_direct), for position
    records and for planes, cooperative and one-CU forms, back-to-back tickets; the f16x2 engine keeps its copies."""
    from bokego_amd import go
    from bokego_amd.engine import LeafEngine
    from bokego_amd.workload import make_batch
    x8, recs = make_batch(256, seed_base=91_000, dtype=np.uint8, with_records=True)
    eng = LeafEngine(weights[0], weights[1], max_batch=512)
    shapes = [(1, 1), (9, 1), (62, 1), (63, 63), (70, 3), (130, 0), (256, 256)]
    def run(kind, B, npol):
        kw = dict(logits=npol > 0, probs=npol > 0, value=True, n_policy=npol)
        return eng.wait(eng.submit_positions(recs[:B], **kw)) if kind == "pos" else eng.eval(x8[:B], **kw)
    eng.set_option("no_direct", 1)
    ref = {(k, *sh): run(k, *sh) for k in ("pos", "planes") for sh in shapes}
    eng.set_option("no_direct", 0)
    for key, want in ref.items():
        got = run(*key)
        for k in want:
            assert np.array_equal(want[k], got[k]), (key, k)
    # four tickets in flight on four slots, then their results in order
    ts = [eng.submit_positions(recs[:B], logits=False, probs=True, value=True, n_policy=1) for B in (62, 17, 80, 5)]
    for t, B in zip(ts, (62, 17, 80, 5)):
        got = eng.wait(t)
        want = ref[("pos", 62, 1)] if B == 62 else None
        if want is not None:
            assert np.array_equal(got["value"], want["value"]) and np.array_equal(got["probs"], want["probs"])
        assert got["value"].shape == (B,) and np.isfinite(got["value"]).all()
    assert eng.stats()["coop_fallbacks"] == 0
    eng.close()


def test_planes_computed_inside_the_leaf_kernel_give_the_same_bits(weights):
    """Round 6: a small request of position records is ONE kernel -- the leaf kernel computes the 27 planes from the 192-byte records
    while it stages them (the feature encoder's device code, bk_encode_dev.h), in every launch form: 12 ... 2 CUs per board, three
    boards on 8 / 4 CUs, whole-board workgroups of one board.  Option no_fuse_encode = 1 puts the encoder kernel back in front:
    the same planes, so every output bit is the same -- and the same as for the reference's own planes of those positions
    (make_batch's planes come from the host encoder, bit-exact against the reference's: tests/test_go_features.py)."""
    from bokego_amd.engine import LeafEngine
    from bokego_amd.workload import make_batch
    x8, recs = make_batch(256, seed_base=93_000, dtype=np.uint8, with_records=True)
    eng = LeafEngine(weights[0], weights[1], max_batch=512)
    shapes = [(1, 1), (3, 3), (12, 1), (17, 17), (25, 25), (35, 35), (45, 45), (62, 1), (80, 80), (81, 0), (110, 1), (200, 7), (256, 256)]
    run = lambda B, npol: eng.wait(eng.submit_positions(recs[:B], logits=npol > 0, probs=npol > 0, value=True, n_policy=npol))  # noqa: E731
    enc0 = eng.stats()["positions_encoded"]
    fused = {sh: run(*sh) for sh in shapes}
    assert eng.stats()["positions_encoded"] - enc0 == sum(B for B, _ in shapes)
    eng.set_option("no_fuse_encode", 1)
    for sh in shapes:
        two = run(*sh)
        planes = eng.eval(x8[:sh[0]], logits=sh[1] > 0, probs=sh[1] > 0, value=True, n_policy=sh[1])
        for k in two:
            assert np.array_equal(fused[sh][k], two[k]) and np.array_equal(fused[sh][k], planes[k]), (sh, k)
    eng.set_option("no_fuse_encode", 0)
    with eng.options(coop=0):                       # ... and in whole-board workgroups of one, two, three boards
        for sh in ((3, 3), (62, 1), (200, 7)):
            got = run(*sh)
            for k in got:
                assert np.array_equal(fused[sh][k], got[k]), (sh, k)
    for nb in (1, 2, 3):
        with eng.options(force_nb=nb):
            got = run(200, 7)
            for k in got:
                assert np.array_equal(fused[(200, 7)][k], got[k]), (nb, k)
    assert eng.stats()["coop_fallbacks"] == 0
    eng.close()
    # the copy-free one-kernel path reaches up to direct_rows = 1,024 records (a 512-game generation's steps); beyond it the three-stream
    # chain with the encoder kernel: the same bits on either side of the limit and with the limit moved
    xb, rb = make_batch(1100, seed_base=94_000, dtype=np.uint8, with_records=True)
    big = LeafEngine(weights[0], weights[1], max_batch=2048)
    want = big.eval(xb, logits=True, probs=True, value=True)
    for B, npol in ((1000, 1000), (1024, 40), (1025, 40), (1100, 0)):
        got = big.wait(big.submit_positions(rb[:B], logits=npol > 0, probs=npol > 0, value=True, n_policy=npol))
        assert np.array_equal(got["value"], want["value"][:B])
        if npol:
            assert np.array_equal(got["logits"], want["logits"][:npol]) and np.array_equal(got["probs"], want["probs"][:npol])
    big.set_option("direct_rows", 0)
    got = big.wait(big.submit_positions(rb[:300], logits=True, probs=True, value=True, n_policy=300))
    assert np.array_equal(got["logits"], want["logits"][:300]) and np.array_equal(got["value"], want["value"][:300])
    big.close()


@pytest.mark.parametrize("precision", ["f32", "f16x2"])
def test_weights_replaced_in_a_live_engine(precision):
    """bk_engine_set_weights (ABI 5): new weights into an engine that has already evaluated -- what an optimizer step or a
    load_state_dict is to the reference's live modules (bin/selfplay.py:80-84,117-119; boke.py:31-37).  Every output bit
    equals a fresh engine's created with those weights, for one net at a time and for both, on the small (cooperative),
    the one-round and the split launch; with a ticket outstanding the call is refused and nothing changes."""
    from bokego_amd.engine import LeafEngine
    sets = _sweep_weight_sets()
    x = np.load(os.path.join(GOLDEN, "sweep_worst.npz"))["features_A"]
    x = np.concatenate([x] * (1 + 900 // len(x)))[:900].astype(np.uint8)
    sizes = (5, 300, 900)

    def run(eng):
        return [eng.eval(x[:B], logits=True, probs=True, value=True) for B in sizes]

    def same(a, b):
        return all(np.array_equal(u[k], v[k]) for u, v in zip(a, b) for k in ("logits", "probs", "value"))

    fresh = {}
    for name, (pw, vw) in sets.items():
        e = LeafEngine(pw, vw, max_batch=1024, precision=precision)
        fresh[name] = run(e)
        e.close()
    e = LeafEngine(pw, vw, max_batch=1024, precision=precision)          # a fresh engine with policy A / value B
    e.set_weights(policy_sd=sets["A"][0])
    mixed = run(e)
    e.close()
    assert not same(fresh["A"], fresh["B"])

    eng = LeafEngine(*sets["A"], max_batch=1024, precision=precision)
    assert same(run(eng), fresh["A"])
    eng.set_weights(value_sd=sets["B"][1])                                # one net at a time ...
    assert same(run(eng), mixed)
    eng.set_weights(policy_sd=sets["B"][0])
    assert same(run(eng), fresh["B"])
    eng.set_weights(*sets["A"])                                           # ... and both
    assert same(run(eng), fresh["A"])
    t = eng.submit(x[:40])
    with pytest.raises(RuntimeError, match="outstanding"):
        eng.set_weights(*sets["B"])
    eng.wait(t)
    assert same(run(eng), fresh["A"])
    eng.close()


def test_load_state_dict_reaches_a_fused_engine():
    """HipPolicyNet.load_state_dict on a net whose engine is shared (nnet.fuse: what the batched MCTS evaluates through):
    the shared engine takes the weights, so a search that holds it plays on with them -- the reference's modules behave
    like that -- instead of the net silently leaving the engine behind."""
    import torch
    from bokego_amd import nnet
    sets = _sweep_weight_sets()
    x = torch.from_numpy(np.load(os.path.join(GOLDEN, "sweep_worst.npz"))["features_A"][:32].astype(np.float32))
    pi, val = nnet.HipPolicyNet(sets["A"][0]), nnet.HipValueNet(sets["A"][1])
    eng = nnet.fuse(pi, val)
    a = pi(x).clone()
    pi.load_state_dict(sets["B"][0])
    assert pi.engine() is eng and val.engine() is eng
    b = pi(x)
    want = nnet.HipPolicyNet(sets["B"][0])(x)
    assert torch.equal(b, want) and not torch.equal(a, b)
    assert torch.equal(torch.from_numpy(eng.eval(x.numpy(), logits=True, probs=False, value=False)["logits"]), want)


@pytest.mark.parametrize("precision", ["f32", "f16x2"])
def test_one_request_of_32768_positions(precision):
    """A request eight times the benchmark's batch (21,846 workgroups, 72 MB of u8 planes): host-buffer and device-resident
    entry points give, bit for bit, what the same positions give 4,096 at a time."""
    import torch
    from bokego_amd.engine import LeafEngine
    from bokego_amd.workload import make_batch
    pw, vw = load_bkw(os.path.join(GOLDEN, "policy_19.bkw")), load_bkw(os.path.join(GOLDEN, "value_synth.bkw"))
    base = make_batch(512, seed_base=777, dtype=np.uint8)
    x = np.tile(base, (64, 1, 1, 1))[np.random.default_rng(1).permutation(32768)]
    eng = LeafEngine(pw, vw, max_batch=32768, precision=precision)
    parts = [eng.eval(x[i:i + 4096], logits=True, probs=True, value=True) for i in range(0, len(x), 4096)]
    ref = {k: np.concatenate([p[k] for p in parts]) for k in ("logits", "probs", "value")}
    big = eng.eval(x, logits=True, probs=True, value=True)
    dev = eng.eval_device(torch.from_numpy(x).cuda(), logits=True, probs=True, value=True)
    torch.cuda.synchronize()
    for k in ref:
        assert np.array_equal(big[k].view(np.uint32), ref[k].view(np.uint32)), k
        assert np.array_equal(dev[k].cpu().numpy().view(np.uint32), ref[k].view(np.uint32)), k
    eng.close()


def test_kernel_order_emulation_matches_the_kernel(weights):
    """tools/emu/kernel_emu.c restates the fp32 kernel's SUMMATION ORDER on the CPU (two chains per conv dot product over the
    halves of the window, MFMA k order inside a chain, four-chain heads) so that tools/error_budget.py can price orders on
    the 49,152-position sweep without a GPU.  That is only worth anything if it is the kernel's order: the logits of 64
    sweep-worst positions must equal the emulation BIT FOR BIT, in every form of the kernel (one request: 3-board workgroups;
    B = 2: two boards; one at a time: the cooperative slices), and the value head's pre-tanh sums give the kernel's values."""
    from bokego_amd.engine import LeafEngine
    chk = np.load(os.path.join(GOLDEN, "emu_check.npz"))
    x = chk["features"]
    eng = LeafEngine(weights[0], weights[1], max_batch=64)
    out = eng.eval(x, logits=True, probs=False, value=True)
    assert np.array_equal(out["logits"], chk["logits_r4"])
    assert not np.array_equal(out["logits"], chk["logits_r3"])           # the one-chain order of rounds 1-3 is a different sum
    assert np.abs(out["value"] - np.tanh(chk["value_pre_tanh_r4"].astype(np.float64))).max() < 1.5e-7
    two = eng.eval(x[:2], logits=True, probs=False, value=True)
    assert np.array_equal(two["logits"], chk["logits_r4"][:2])
    for i in (0, 17, 63):
        one = eng.eval(x[i:i + 1], logits=True, probs=False, value=True)
        assert np.array_equal(one["logits"], chk["logits_r4"][i:i + 1]) and one["value"][0] == out["value"][i]
    eng.close()
