"""CPU-only: the C-ABI library loads, exports every symbol include/bokego_amd.h declares, and fails
loudly (no CPU fallback) when there is no GPU.  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, REPO


def _header_functions(name="bokego_amd.h"):
    src = open(os.path.join(REPO, "include", name)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bk_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(REPO, "bokego_amd", "libbokego_amd.so")):
        g.build()
    from bokego_amd import _lib
    return _lib.load()


def test_header_symbols_exported(lib):
    from bokego_amd import _lib
    fns = _header_functions()
    assert len(fns) >= 14
    for f in fns:
        assert hasattr(lib, f), f"{f} declared in include/bokego_amd.h but not exported"
        assert f in _lib.SYMBOLS, f"{f} has no ctypes prototype in bokego_amd/_lib.py"
    assert set(_lib.SYMBOLS) == set(fns)
    assert lib.bk_abi_version() == 7
    assert lib.bk_has_test_hooks() == 0 and not hasattr(lib, "bk_debug_fail_nth_hip_call")    # the shipped build carries no test code


def test_plan_flops_counts_the_tile_tables(lib):
    """bk_plan_flops (no GPU): executed fp32-MFMA FLOP of a request from the compiled kernel's tile tables, against the
    algorithmic FLOP of SURVEY 8d.  The 3-board figure is the one the rocprofv3 counters give per workgroup
    (profiles/r02_pmc_f32.json: SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 / (SQ_WAVES / 8) = 435,355,648)."""
    def q(p, v, coop=0):
        ex, al, nl = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
        assert lib.bk_plan_flops(p, v, 256, coop, ctypes.byref(ex), ctypes.byref(al), ctypes.byref(nl)) == 0
        return ex.value, al.value, nl.value
    ex, al, nl = q(768, 0)                                  # one round of 3-board workgroups, PolicyNet only
    assert nl == 1 and ex / 256 == 2048 * 8 * 2 * (170 * 7 + 6 * 63 * 32) == 435355648 and al == 768 * 2 * 66706944
    ex, al, nl = q(4096, 4096)                              # the bench step: 10 rounds of 3-board workgroups + a 2-board tail
    assert nl == 2 and al == 4096 * 266838272
    assert ex == 2560 * 435355648 + 256 * 2048 * 8 * (235 * 7 + 6 * 87 * 32)
    assert 1.05 < ex / al < 1.10
    one = 2048 * 8 * (130 * 7 + 6 * 48 * 32)                # a single board: 6 tiles, the two y-edge tiles skip their outward taps
    every = 2048 * 8 * 6 * (25 * 7 + 6 * 9 * 32)            # ... every tap of every tile (round 3; still the 3-, 8- and 12-CU cooperative forms)
    assert q(1, 62, coop=1) == (63 * one, 2.0 * (66706944 + 62 * 66712192), 1) == q(1, 62, coop=0)     # 63 tasks: 4 CUs per board
    assert q(1, 70, coop=1)[0] == 71 * every and q(1, 70, coop=0)[0] == 71 * one                       # 71 tasks: 3 CUs per board
    assert one / every < 0.9
    assert q(2, 150, coop=1)[0] == (1 + 50) * 435355648 and q(20, 300, coop=1)[0] == (7 + 100) * 435355648   # three boards on 4 / 2 CUs: the 3-board tile set per group
    assert q(2, 150, coop=0)[0] == 152 * one
    assert q(0, 0) == (0.0, 0.0, 0) and lib.bk_plan_flops(-1, 0, 256, 0, None, None, None) == -1
    assert lib.bk_plan_flops(5, 5, 256, 0, None, None, None) == 0   # every out pointer may be NULL


def test_struct_layout_matches_header():
    from bokego_amd import _lib
    # 7*6 + 2 pointers in the trunk, +12 in the value head (include/bokego_amd.h)
    assert ctypes.sizeof(_lib.TrunkWeights) == 44 * 8
    assert ctypes.sizeof(_lib.ValueWeights) == 56 * 8
    assert ctypes.sizeof(_lib.Stats) == 96 + 5 * 8       # ABI 6: mean_batch, queue_wait_ms_sum, queue_wait_count, host_wait_ms_sum, failed_submissions


def test_create_argument_errors(lib):
    h = ctypes.c_void_p()
    assert lib.bk_engine_create(None, None, 0, 16, ctypes.byref(h)) == -1      # BK_ERR_ARG
    assert b"policy/value" in lib.bk_last_error(None)
    from bokego_amd import _lib
    pw = _lib.PolicyWeights()                                                  # all NULL pointers
    assert lib.bk_engine_create(ctypes.byref(pw), None, 0, 16, ctypes.byref(h)) == -1
    assert lib.bk_engine_create(ctypes.byref(pw), None, 0, 0, ctypes.byref(h)) == -1
    assert lib.bk_engine_destroy(None) == -1
    assert lib.bk_engine_max_batch(None) == -1


def test_no_cpu_fallback_without_gpu(lib):
    """On a box without a GPU the engine must refuse to exist (BK_ERR_NO_GPU), not fall back."""
    if lib.bk_device_count() > 0:
        pytest.skip("a GPU is present")
    from bokego_amd.bkw import load_bkw
    from bokego_amd.engine import LeafEngine
    with pytest.raises(RuntimeError, match="BK_ERR_NO_GPU"):
        LeafEngine(load_bkw(os.path.join(GOLDEN, "policy_19.bkw")), None)


def test_state_dict_validation(lib):
    from bokego_amd.bkw import load_bkw
    from bokego_amd.engine import LeafEngine
    sd = load_bkw(os.path.join(GOLDEN, "policy_19.bkw"))
    bad = dict(sd)
    del bad["conv.9.weight"]
    with pytest.raises(KeyError):
        LeafEngine(bad, None)
    bad = dict(sd)
    bad["conv.0.weight"] = np.zeros((128, 27, 3, 3), np.float32)
    with pytest.raises(ValueError):
        LeafEngine(bad, None)
    with pytest.raises(TypeError):
        LeafEngine(None, None)


def test_bkw_roundtrip(tmp_path):
    import torch
    from bokego_amd.bkw import load_bkw, save_bkw, state_dict_to_tensors, tensors_to_state_dict
    sd = load_bkw(os.path.join(GOLDEN, "value_synth.bkw"))
    assert sd["conv.0.weight"].shape == (128, 27, 5, 5) and sd["conv.21.bias"].shape == (1, 9, 9)
    assert sd["lin1.weight"].shape == (64, 81) and sd["bn.running_var"].shape == (1,)
    p = tmp_path / "x.bkw"
    save_bkw(str(p), sd)
    back = load_bkw(str(p))
    assert list(back) == list(sd) and all(np.array_equal(back[k], sd[k]) for k in sd)
    tsd = tensors_to_state_dict(sd)
    assert "conv.1.num_batches_tracked" in tsd and tsd["conv.3.weight"].dtype == torch.float32
    again = state_dict_to_tensors(tsd)
    assert all(np.array_equal(again[k], sd[k]) for k in sd)


def test_host_library_exports_go_and_tree_headers():
    """libbkgo.so (board, features, tree pool) exports everything bokego_go.h / bokego_tree.h declare."""
    import ctypes as C
    from bokego_amd import go, selfplay
    lib = go.golib()
    selfplay.treelib()
    for hdr, table in (("bokego_go.h", go.GO_SYMBOLS), ("bokego_tree.h", selfplay.TREE_SYMBOLS)):
        fns = _header_functions(hdr)
        assert fns and set(fns) == set(table), (hdr, set(fns) ^ set(table))
        for f in fns:
            assert hasattr(lib, f)
    assert lib.bk_go_abi_version() == 6
    assert C.sizeof(selfplay.GameStats) == 81 * 8 + 24 and C.sizeof(selfplay.EvaluatorStruct) == 24 and C.sizeof(selfplay.RunInfo) == 40
    assert C.sizeof(go.Pos) == 192 and C.sizeof(selfplay.SearchParams) == 104 and C.sizeof(selfplay.NodeInfo) == 24 and C.sizeof(selfplay.GameInfo) == 56


def test_lds_edge_tables_match_generator():
    """The edge-tile row tables compiled into bk_kernels_f16.hip are the ones tools/lds_layout.py generates, and the
    generator's bank model says every ds_read_b128 lane group is conflict-free with them."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("lds_layout", os.path.join(REPO, "tools", "lds_layout.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    src = open(os.path.join(REPO, "bokego_amd", "csrc", "bk_kernels_f16.hip")).read()

    def table(name):
        body = re.search(name + r"\[2\]\[32\] = \{(.*?)\};", src, flags=re.S).group(1)
        rows = re.findall(r"\{([^{}]*)\}", body)
        return [[int(v) for v in r.split(",")] for r in rows]

    B, X, K = table("kEdgeB"), table("kEdgeX"), table("kEdgeKey")
    for e, y in enumerate((0, 8)):
        lanes, keys = m.edge_tile(y)
        assert B[e] == [it[0] if it else -1 for it in lanes]
        assert X[e] == [it[1] if it else 0 for it in lanes]
        assert K[e] == keys
    assert m.cycles(m.new_rows()) == 1.0 and m.cycles(m.old_rows()) > 1.3


def test_comm_library_symbols_and_cpu_errors():
    """libbkcomm.so (RCCL all-reduce of the self-play statistics) exports what include/bokego_comm.h declares and
    refuses to create a communicator without a GPU."""
    from bokego_amd import comm
    lib = comm.commlib()
    fns = _header_functions("bokego_comm.h")
    assert set(fns) == set(comm.COMM_SYMBOLS)
    for f in fns:
        assert hasattr(lib, f)
    assert lib.bk_comm_abi_version() == 2
    import torch
    if not torch.cuda.is_available():
        h = ctypes.c_void_p()
        ident = (ctypes.c_uint8 * 128)()
        assert lib.bk_comm_init(0, 1, ident, 0, ctypes.byref(h)) != 0
        assert b"device_id" in lib.bk_comm_last_error()
        assert lib.bk_comm_init(2, 1, ident, 0, ctypes.byref(h)) != 0       # rank >= world
        assert lib.bk_comm_allreduce_sum_f64(None, None, 3) != 0


def test_launch_planner_choices(lib, monkeypatch):
    """bk_plan_query: the planner's decisions as a pure function -- CUs per board of the cooperative small-batch form by
    task count (what fits is ceil(tasks / 8) groups on the n_cu / 8 CUs of an XCD) and boards per workgroup otherwise."""
    # a pure function of its arguments: the environment is read by bk_engine_create only -- variables
    # that used to switch the planner must not move it
    for v in ("BK_COOP", "BK_COOP3", "BK_FORCE_NB"):
        monkeypatch.setenv(v, "0")
    nb = ctypes.c_int(0)
    q = lambda npol, nval, prec=0, n_cu=256: lib.bk_plan_query(npol, nval, n_cu, prec, ctypes.byref(nb))  # noqa: E731
    # (12 CUs per board while at most two boards share an XCD -- up to 16 tasks, round 6 -- then 8, 6, 4, 3, 2)
    assert [q(1, 1), q(1, 7), q(1, 8), q(1, 15), q(1, 16), q(1, 31), q(1, 32), q(1, 39), q(1, 40), q(1, 63), q(1, 64), q(1, 79), q(1, 80),
            q(0, 128), q(0, 0)] == [12, 12, 12, 12, 8, 8, 6, 6, 4, 4, 3, 3, 108, 2, 0]
    # 81..96 tasks in at most 32 groups of three boards of one net: three boards on EIGHT CUs (code 108; round 5: 140 us against
    # the 2-CUs-per-board form's 174); one group too many, or one task, and the 2-CUs-per-board form runs
    assert [q(0, 81), q(3, 93), q(0, 96), q(6, 90), q(1, 95), q(4, 92), q(0, 97), q(1, 79)] == [108, 108, 108, 108, 2, 2, 2, 3]
    # between the whole-board forms' ranges: three boards of one net on 4 CUs (code 104: 129..192 tasks, while the groups
    # of three fit 8 to an XCD) and on 2 CUs (102: 257..384 tasks, 16 groups to an XCD)
    assert [q(1, 128), q(2, 150), q(6, 186), q(1, 190), q(1, 191), q(8, 200), q(0, 256), q(1, 256), q(20, 300), q(30, 340), q(3, 381),
            q(4, 380), q(14, 370), q(1, 384)] == [104, 104, 104, 0, 0, 0, 0, 102, 102, 102, 102, 0, 0, 0]
    assert q(1, 62, 1) == 0                      # f16x2 engines keep the one-CU form
    assert q(1, 15, 0, 64) == 4 and q(1, 16, 0, 64) == 2 and q(1, 32, 0, 64) == 0   # a 64-CU device: 8 CUs per XCD
    assert q(4096, 4096) == 0 and nb.value == 3
    assert q(100, 100) == 0 and nb.value == 1
    assert q(300, 300) == 0 and nb.value in (2, 3)
    assert lib.bk_plan_query(-1, 0, 256, 0, None) == -1 and lib.bk_plan_query(1, 1, 0, 0, None) == -1


def test_the_request_path_does_not_read_the_environment():
    """getenv is not safe against a concurrent setenv (os.environ[...] = in another Python thread), and a
    stray BK_* variable must not change a running engine: the engine sources read the environment in ONE helper, called from
    bk_engine_create; the kernels' file not at all."""
    src = open(os.path.join(REPO, "bokego_amd", "csrc", "bk_engine.cpp")).read()
    code = re.sub(r"//[^\n]*", "", src)
    assert len(re.findall(r"\bgetenv\s*\(", code)) == 1
    create = code[code.index("int bk_engine_create("):code.index("int bk_engine_destroy(")]
    uses = [m.start() for m in re.finditer(r"\benv_(?:str|int)\s*\(", code)]
    inside = [u for u in uses if code.index("int bk_engine_create(") <= u < code.index("int bk_engine_destroy(")]
    helpers = [u for u in uses if u < code.index("struct bk_engine {")]          # the helper's own definition + the roctx loader
    assert len(inside) >= 8 and len(inside) + len(helpers) == len(uses) and "env_str" in create
    # ... nor do the tree / board library's (BK_NO_LANES is now the pool switch bk_pool_set_lanes) or the collective's
    for f in ("bk_kernels.hip", "bk_kernels_f16.hip", "bk_encode.hip", "bk_tree.cpp", "bk_go.cpp", "bk_comm.cpp"):
        code = re.sub(r"//[^\n]*", "", open(os.path.join(REPO, "bokego_amd", "csrc", f)).read())
        assert not re.search(r"\bgetenv\s*\(", code), f
