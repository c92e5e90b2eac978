"""-m gpu: what needs fault injection runs on the -DBK_TEST_HOOKS build of the engine library (make hooks ->
bokego_amd/libbokego_amd_hooks.so), loaded BESIDE the shipped one: the shipped libbokego_amd.so carries no test code -- no
`coop_fault` branch in its kernels, no fault counter in its HIP-call wrapper, no bk_debug_* export.  The
two builds are the same sources; `test_hook_build_gives_the_shipped_bits` compares their outputs."""
import ctypes
import os

import numpy as np
import pytest

from bokego_amd import _lib as L
from bokego_amd.bkw import load_bkw

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
HOOKS = L.HOOKS_LIB_PATH


@pytest.fixture(scope="module", autouse=True)
def hooks_library():
    """The fault-injection build is the tests' own: normally it travels with the tree (`make all` builds it beside the shipped
    library); where it is missing it is built here (`make hooks`, ~30 s), and a build without its hooks fails the module."""
    import subprocess
    if not os.path.exists(HOOKS):
        subprocess.check_call(["make", "-C", os.path.join(os.path.dirname(HOOKS), "csrc"), "hooks"])
    assert L.load(HOOKS).bk_has_test_hooks() == 1, "libbokego_amd_hooks.so was built without -DBK_TEST_HOOKS"


@pytest.fixture(scope="module")
def weights():
    return load_bkw(os.path.join(GOLDEN, "policy_19.bkw")), load_bkw(os.path.join(GOLDEN, "value_synth.bkw"))


def _same(a, b):
    return all(np.array_equal(a[k], b[k]) for k in a)


def _hooked(pw, vw, **kw):
    from bokego_amd.engine import LeafEngine
    eng = LeafEngine(pw, vw, lib_path=HOOKS, **kw)
    assert eng._lib.bk_has_test_hooks() == 1
    return eng


def test_shipped_library_has_no_test_hooks(weights):
    """nm-level: no bk_debug_* symbol, bk_has_test_hooks() == 0, and the hook options are unknown names to it."""
    from bokego_amd.engine import LeafEngine
    lib = L.load()
    assert lib.bk_has_test_hooks() == 0 and not hasattr(lib, "bk_debug_fail_nth_hip_call")
    eng = LeafEngine(weights[0], weights[1], max_batch=64)
    for name in ("coop_fault", "fault_submit"):
        with pytest.raises(ValueError, match="unknown engine option"):
            eng.set_option(name, 1)
    eng.close()


def test_hook_build_gives_the_shipped_bits(weights):
    from bokego_amd.engine import LeafEngine
    from bokego_amd.workload import make_batch
    x = make_batch(400, seed_base=31_000, dtype=np.uint8)
    a, b = LeafEngine(weights[0], weights[1], max_batch=512), _hooked(weights[0], weights[1], max_batch=512)
    for B, npol in ((1, 1), (62, 1), (100, 28), (150, 2), (340, 30), (400, 400)):
        assert _same(a.eval(x[:B], logits=True, probs=True, value=True, n_policy=npol),
                     b.eval(x[:B], logits=True, probs=True, value=True, n_policy=npol)), (B, npol)
    a.close()
    b.close()


def test_cooperative_form_falls_back_when_a_peer_never_arrives():
    """The option coop_fault (hook build) makes one slice of one board leave before a meeting point: its peers give up after the bounded
    wait and raise the flag that travels with the outputs, bk_wait redoes the request with one CU per board
    (bk_stats().coop_fallbacks), the
    outputs are the usual bits, and the next cooperative launch finds its counters clean."""
    from bokego_amd.engine import LeafEngine
    from bokego_amd.workload import make_batch
    x = make_batch(48, seed_base=5_000, dtype=np.uint8)
    pw, vw = load_bkw(os.path.join(GOLDEN, "policy_19.bkw")), load_bkw(os.path.join(GOLDEN, "value_synth.bkw"))
    eng = _hooked(pw, vw, max_batch=64)
    eng.set_option("coop", 0)
    ref = eng.eval(x, logits=True, probs=True, value=True, n_policy=5)
    eng.set_option("coop", -1)
    eng.set_option("coop_fault", 1)
    bad = eng.eval(x, logits=True, probs=True, value=True, n_policy=5)
    eng.set_option("coop_fault", 0)
    st = eng.stats()
    assert st["coop_fallbacks"] == 1 and st["coop_launches"] == 1
    good = eng.eval(x, logits=True, probs=True, value=True, n_policy=5)
    st = eng.stats()
    assert st["coop_fallbacks"] == 1 and st["coop_launches"] == 2
    for k in ref:
        assert np.array_equal(ref[k], bad[k]) and np.array_equal(ref[k], good[k]), k
    eng.close()


def test_a_failed_cooperative_request_of_position_records_is_redone_from_the_records():
    """Round 6: a small request of position records has no planes in memory (the leaf kernel computes them while staging); when a
    peer deserts, bk_wait's redo with one CU per board reads the same records again: the usual bits."""
    from bokego_amd.workload import make_batch
    x8, recs = make_batch(64, seed_base=8_000, dtype=np.uint8, with_records=True)
    pw, vw = load_bkw(os.path.join(GOLDEN, "policy_19.bkw")), load_bkw(os.path.join(GOLDEN, "value_synth.bkw"))
    eng = _hooked(pw, vw, max_batch=64)
    ref = eng.eval(x8[:48], logits=True, probs=True, value=True, n_policy=5)
    eng.set_option("coop_fault", 1)
    bad = eng.wait(eng.submit_positions(recs[:48], logits=True, probs=True, value=True, n_policy=5))
    eng.set_option("coop_fault", 0)
    st = eng.stats()
    assert st["coop_fallbacks"] == 1
    good = eng.wait(eng.submit_positions(recs[:48], logits=True, probs=True, value=True, n_policy=5))
    assert eng.stats()["coop_fallbacks"] == 1
    for k in ref:
        assert np.array_equal(ref[k], bad[k]) and np.array_equal(ref[k], good[k]), k
    eng.close()


def test_cooperative_failure_is_sticky_for_requests_queued_behind_it():
    """A is submitted with a deserting slice, B and C right behind it, all three cooperative, before
    anything is waited for.  A's peers time out and raise the engine's poison word; B and C run before the host has
    cleared the counters, see the word at entry, flag themselves and are redone by bk_wait as well: nobody hands out
    results computed on stale counters.  D, submitted after the waits, is an ordinary clean cooperative launch."""
    from bokego_amd.engine import LeafEngine
    from bokego_amd.workload import make_batch
    xs = [make_batch(n, seed_base=6_000 + 100 * i, dtype=np.uint8) for i, n in enumerate((48, 30, 62, 17))]
    pw, vw = load_bkw(os.path.join(GOLDEN, "policy_19.bkw")), load_bkw(os.path.join(GOLDEN, "value_synth.bkw"))
    eng = _hooked(pw, vw, max_batch=64)
    eng.set_option("coop", 0)
    refs = [eng.eval(x, logits=True, probs=True, value=True, n_policy=3) for x in xs]
    eng.set_option("coop", -1)
    eng.set_option("coop_fault", 1)
    tA = eng.submit(xs[0], logits=True, probs=True, value=True, n_policy=3)
    eng.set_option("coop_fault", 0)
    tB = eng.submit(xs[1], logits=True, probs=True, value=True, n_policy=3)
    tC = eng.submit(xs[2], logits=True, probs=True, value=True, n_policy=3)
    outs = [eng.wait(tA), eng.wait(tB), eng.wait(tC)]
    st = eng.stats()
    assert st["coop_launches"] == 3 and st["coop_fallbacks"] == 3
    outs.append(eng.eval(xs[3], logits=True, probs=True, value=True, n_policy=3))
    st = eng.stats()
    assert st["coop_launches"] == 4 and st["coop_fallbacks"] == 3
    for ref, out in zip(refs, outs):
        for k in ref:
            assert np.array_equal(ref[k], out[k]), k
    eng.close()


def test_three_boards_form_falls_back_when_a_peer_never_arrives():
    """The three-boards-on-2/4-CUs forms (tests/test_gpu_parity.py::test_three_boards_on_two_or_four_cus_give_the_same_bits)
    with a deserting peer: the usual fallback, the usual bits, clean counters afterwards."""
    from bokego_amd.workload import make_batch
    x = make_batch(384, seed_base=77_000, dtype=np.uint8)
    pw, vw = load_bkw(os.path.join(GOLDEN, "policy_19.bkw")), load_bkw(os.path.join(GOLDEN, "value_synth.bkw"))
    eng = _hooked(pw, vw, max_batch=512)
    eng.set_option("coop3", 0)
    ref = {sh: eng.eval(x[:sh[0]], logits=True, probs=True, value=True, n_policy=sh[1]) for sh in ((150, 2), (340, 30))}
    assert eng.stats()["coop_launches"] == 0
    eng.set_option("coop3", -1)
    c0 = eng.stats()["coop_launches"]
    eng.set_option("coop_fault", 1)
    bad = [eng.eval(x[:B], logits=True, probs=True, value=True, n_policy=npol) for B, npol in ((150, 2), (340, 30))]
    eng.set_option("coop_fault", 0)
    st = eng.stats()
    assert st["coop_fallbacks"] == 2 and st["coop_launches"] - c0 == 2
    good = eng.eval(x[:340], logits=True, probs=True, value=True, n_policy=30)
    assert eng.stats()["coop_fallbacks"] == 2
    for k in good:
        assert np.array_equal(ref[(150, 2)][k], bad[0][k]) and np.array_equal(ref[(340, 30)][k], bad[1][k]) and np.array_equal(ref[(340, 30)][k], good[k]), k
    # three boards on eight CUs (forced): the same fallback
    with eng.options(coop=0):
        ref8 = eng.eval(x[:88], logits=True, probs=True, value=True, n_policy=2)
    eng.set_option("coop3", 8)
    eng.set_option("coop_fault", 1)
    bad8 = eng.eval(x[:88], logits=True, probs=True, value=True, n_policy=2)
    eng.set_option("coop_fault", 0)
    assert eng.stats()["coop_fallbacks"] == 3
    good8 = eng.eval(x[:88], logits=True, probs=True, value=True, n_policy=2)
    assert eng.stats()["coop_fallbacks"] == 3
    assert all(np.array_equal(ref8[k], bad8[k]) and np.array_equal(ref8[k], good8[k]) for k in ref8)
    eng.close()


@pytest.mark.parametrize("precision", ["f32", "f16x2"])
def test_a_submission_that_fails_part_way_leaves_a_clean_slot(weights, precision):
    """The option fault_submit = n (hook build) makes the n-th HIP call of a ticket submission report a failure (the call is not made): wherever
    that lands -- a copy, the encoder, an event hop of the three-stream chain, the kernel launch, the copy back -- the call
    returns BK_ERR_HIP, nothing stays in flight on the slot it had taken, and the next requests on the same engine (the same
    slot among them) give the usual bits.  Small (single-stream, copy-free), mid-size (three streams) and two-part requests,
    planes and position records."""
    from bokego_amd.engine import LeafEngine
    from bokego_amd.workload import make_batch
    x8, recs = make_batch(2400, seed_base=77_000, dtype=np.uint8, with_records=True)
    xf = x8.astype(np.float32)
    eng = _hooked(weights[0], weights[1], max_batch=2400, precision=precision)
    kw = dict(logits=True, probs=True, value=True)
    cases = {"small_pos": lambda: eng.wait(eng.submit_positions(recs[:62], n_policy=1, **kw)),
             "small_planes": lambda: eng.eval(x8[:40], **kw),
             "mid_pos": lambda: eng.wait(eng.submit_positions(recs[:900], **kw)),
             "mid_planes": lambda: eng.eval(x8[:900], **kw),
             "two_part_f32": lambda: eng.eval(xf[:2400], **kw)}
    ref = {k: f() for k, f in cases.items()}
    failures = 0
    for name, f in cases.items():
        for n in range(1, 40):
            eng.set_option("fault_submit", n)
            try:
                out = f()
            except RuntimeError as ex:
                assert "BK_ERR_HIP" in str(ex), ex
                failures += 1
                eng._pending.clear()
                eng.set_option("fault_submit", 0)
                for k2 in ("small_pos", "mid_planes", name, "small_planes"):      # four requests behind the failed one
                    assert _same(cases[k2](), ref[k2]), (name, n, k2)
                continue
            finally:
                eng.set_option("fault_submit", 0)
            assert _same(out, ref[name])       # n beyond the submission's last HIP call: nothing was injected
            break
        else:
            pytest.fail(f"{name}: more than 39 HIP calls in one submission?")
    st = eng.stats()
    assert failures >= 5 * 3 and st["failed_submissions"] == failures
    assert st["coop_fallbacks"] == 0
    eng.close()


def test_new_weights_arrive_all_or_not_at_all(weights):
    """bk_engine_set_weights packs everything on the host, uploads into FRESH device buffers and switches the engine over
    only when every buffer is in place (a failure part-way used to leave a mixture of old and new weights in the
    buffers the kernels read).  An upload that fails -- injected through the same hook, armed for one call -- leaves every
    output bit as before; the call also waits for device-path launches on caller streams before the old buffers go."""
    import ctypes
    import torch
    from bokego_amd.engine import LeafEngine
    x = np.load(os.path.join(GOLDEN, "features.npz"))["incremental"][:300].astype(np.uint8)
    eng = _hooked(weights[0], weights[1], max_batch=512)
    kw = dict(logits=True, probs=True, value=True)
    before = [eng.eval(x[:B], **kw) for B in (5, 300)]
    other_p = {k: v for k, v in weights[1].items() if k.startswith("conv.")}     # the policy_17 trunk + head (value_synth carries it)
    lib = eng._lib
    lib.bk_debug_fail_nth_hip_call.argtypes = [ctypes.c_void_p, ctypes.c_int]
    for n in (2, 5, 9, 12):                               # at four points of the six uploads (hipMalloc + hipMemcpy each)
        lib.bk_debug_fail_nth_hip_call(eng._h, n + 2)     # + hipSetDevice + hipDeviceSynchronize in front
        with pytest.raises(RuntimeError, match="BK_ERR_HIP"):
            eng.set_weights(policy_sd=other_p)
        after = [eng.eval(x[:B], **kw) for B in (5, 300)]
        assert all(_same(a, b) for a, b in zip(before, after)), n
    # a device-path launch on a non-blocking caller stream is still running when the weights are replaced
    s = torch.cuda.Stream()
    d = torch.from_numpy(np.concatenate([x] * 8)[:512]).cuda()
    with torch.cuda.stream(s):
        for _ in range(6):
            o = eng.eval_device(d, logits=True, probs=False, value=False)
    eng.set_weights(policy_sd=other_p)
    s.synchronize()
    want = eng.eval(x[:5], **kw)
    assert not np.array_equal(want["logits"], before[0]["logits"])
    ref = _hooked(weights[0], weights[1], max_batch=512)
    got = ref.eval(np.concatenate([x] * 8)[:512], logits=True, probs=False, value=False)
    assert np.array_equal(o["logits"].cpu().numpy(), got["logits"])     # the launches in flight finished on the OLD weights
    ref.close()
    eng.close()
