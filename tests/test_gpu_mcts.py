"""-m gpu: drop-in nets + batched MCTS on the HIP engine (BASELINE configs 1 and 3)."""
import json
import os
import time

import numpy as np
import pytest
import torch

from bokego_amd.bkw import load_bkw

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sds():
    return load_bkw(os.path.join(GOLDEN, "policy_19.bkw")), load_bkw(os.path.join(GOLDEN, "value_synth.bkw"))


@pytest.fixture(scope="module")
def gold():
    return (np.load(os.path.join(GOLDEN, "features.npz"))["incremental"].astype(np.float32),
            np.load(os.path.join(GOLDEN, "nets.npz")))


def test_policy_prefix_split(sds, gold):
    from bokego_amd.engine import LeafEngine
    f, n = gold
    e = LeafEngine(sds[0], sds[1], max_batch=600)
    full = e.eval(f, logits=True, probs=True, value=True)
    for k in (0, 1, 2, 5, 13, 100, 536):
        o = e.eval(f, logits=True, probs=True, value=True, n_policy=k)
        assert o["logits"].shape == (k, 81) and o["value"].shape == (536,)
        assert np.array_equal(o["logits"], full["logits"][:k]) and np.array_equal(o["probs"], full["probs"][:k])
        assert np.array_equal(o["value"], full["value"])
    o = e.eval(f[:7], logits=False, probs=True, value=False, n_policy=3)
    assert np.array_equal(o["probs"], full["probs"][:3]) and "value" not in o
    with pytest.raises(ValueError):
        e.eval(f[:7], n_policy=8)
    e.close()


def test_dropin_nets_reference_surface(sds, gold):
    """What boke.py:30-38 and nnet.py:265-297 do with the nets, on the HIP shims."""
    from bokego_amd import go, nnet
    f, n = gold
    pi = nnet.HipPolicyNet()
    with pytest.raises(RuntimeError):
        pi(torch.zeros(1, 27, 9, 9))                         # no weights yet
    pi.load_state_dict({k: torch.from_numpy(v) for k, v in sds[0].items()})
    assert pi.eval() is pi and pi.to(torch.device("cpu")) is pi
    val = nnet.HipValueNet()
    val.load_state_dict(sds[1])
    x = torch.from_numpy(f[:9])
    lg, v = pi(x), val(x)
    assert lg.shape == (9, 81) and v.shape == (9, 1) and lg.dtype == torch.float32
    assert np.abs(lg.numpy() - n["logits_b1"][:9]).max() < 1e-4
    assert np.abs(v.numpy().reshape(-1) - n["values_b1"][:9]).max() < 1e-4
    # cuda tensors in -> cuda tensors out
    assert pi(x.cuda()).is_cuda and np.abs(pi(x.cuda()).cpu().numpy() - lg.numpy()).max() == 0
    g = go.Game()
    d = nnet.policy_dist(pi, g)
    assert abs(d.probs[40].item() - 0.818645) < 1e-5 and d.probs.argmax().item() == 40   # SURVEY 8c known answer
    assert abs(nnet.value(val, g) - float(n["values_b1"][0])) < 1e-4
    torch.manual_seed(0)
    assert 0 <= nnet.policy_sample(pi, g).item() < 81
    # ValueNet.load_policy_dict overlays a policy trunk (nnet.py:103-107)
    v2 = nnet.HipValueNet(sds[1])
    v2.load_policy_dict(pi.state_dict())
    assert np.array_equal(v2.state_dict()["conv.3.weight"].numpy(), sds[0]["conv.3.weight"])
    assert abs(v2(x)[0, 0].item() - v[0, 0].item()) > 1e-6
    eng = nnet.fuse(pi, val)
    assert pi.engine() is eng and val.engine() is eng
    assert np.abs(pi(x).numpy() - lg.numpy()).max() == 0


def test_config3_genmove_1600_rollouts_matches_reference(sds):
    """BASELINE config 3: 1600 rollouts/move, batched leaf queue, 1 GPU: same moves and the same
    root-child visit counts as the trace recorded from the reference (tests/golden/mcts_trace.json)."""
    from bokego_amd import nnet
    from bokego_amd.mcts import MCTS, Go_MCTS
    t = json.load(open(os.path.join(GOLDEN, "mcts_trace.json")))["r1600"]
    pi, val = nnet.HipPolicyNet(sds[0]), nnet.HipValueNet(sds[1])
    torch.manual_seed(0)
    tree = MCTS(Go_MCTS(), pi, val, no_sim=True)
    t0 = time.time()
    for ref in t["moves"]:
        tree.rollout(t["rollouts"])
        kids = {c.mv: tree.N[c] for c in tree.children[tree.root]}
        rootN = tree.N[tree.root]
        best = tree.choose()
        assert best.last_move == ref["move"], (best.last_move, ref["alpha"])
        assert kids == {int(k): v for k, v in ref["child_N"].items()}
        assert rootN == ref["root_N"]
    dt = (time.time() - t0) / len(t["moves"])
    ev = tree.evaluator
    print(f"\nconfig3: {dt*1e3:.0f} ms/move, {ev.n_positions/len(t['moves']):.0f} evals/move, "
          f"mean batch {ev.n_positions/ev.n_batches:.1f}, reference evals: {t['n_value_evals']} value / {t['n_policy_evals']} policy")
    assert ev.n_positions / ev.n_batches > 20


def test_gtp_session_matches_reference_transcript_on_gpu(sds):
    """The scripted GTP session recorded from the reference (200 rollouts per genmove), on the HIP nets."""
    from bokego_amd import nnet
    from bokego_amd.gtp import GTP
    from bokego_amd.mcts import Go_MCTS
    t = json.load(open(os.path.join(GOLDEN, "gtp_transcript.json")))
    g = GTP(Go_MCTS(), nnet.HipPolicyNet(sds[0]), nnet.HipValueNet(sds[1]), no_sim=True, time_lim=None,
            n_rollouts=t["n_rollouts"])
    g.running = True
    for cmd, want in t["session"]:
        assert g.send(cmd) == want, cmd


def test_config3_on_native_tree(sds):
    """Same BASELINE config 3 trace (1600 rollouts x 10 moves) with the tree in C++ (NativeMCTS)."""
    from bokego_amd import nnet
    from bokego_amd.mcts_native import NativeMCTS, Position
    t = json.load(open(os.path.join(GOLDEN, "mcts_trace.json")))["r1600"]
    tree = NativeMCTS(Position(), nnet.HipPolicyNet(sds[0]), nnet.HipValueNet(sds[1]))
    t0 = time.time()
    for ref in t["moves"]:
        tree.rollout(t["rollouts"])
        kids = {m: n for m, (n, _) in tree.child_stats().items()}
        assert kids == {int(k): v for k, v in ref["child_N"].items()}
        assert abs(tree.winrate() - ref["root_winrate"]) < 1e-4
        assert tree.choose().last_move == ref["move"]
    dt = (time.time() - t0) / len(t["moves"])
    ev = tree.evaluator
    print(f"\nconfig3 native: {dt*1e3:.1f} ms/move, mean batch {ev.positions/ev.batches:.1f}")


def _replay_whole_game(tree, t, native):
    """rollout(1600) + choose() for every move of the recorded game; returns seconds per move"""
    t0 = time.time()
    for ply, ref in enumerate(t["moves"]):
        tree.rollout(t["rollouts"])
        if native:
            kids = {m: n for m, (n, _) in tree.child_stats().items()}
            root_n = tree.N[tree.root]
        else:
            kids = {c.mv: tree.N[c] for c in tree.children[tree.root]}
            root_n = tree.N[tree.root]
        assert kids == {int(k): v for k, v in ref["child_N"].items()}, ply
        assert root_n == ref["root_N"], ply
        assert abs(tree.winrate() - ref["root_winrate"]) < 1e-4, ply
        best = tree.choose()
        assert best.last_move == ref["move"], (ply, best.last_move, ref["alpha"])
    assert tree.root.board == t["final_board"] and tree.root.turn == t["final_turn"]
    assert tree.root._terminal and tree.choose().key() == tree.root.key()      # turn > MAX_TURNS: mcts.py:116-118,362-364
    return (time.time() - t0) / len(t["moves"])


def test_config3_whole_game_matches_reference_on_both_trees(sds):
    """VERDICT r4 next #1b: the regime in which ms/move is quoted -- a WHOLE game (81 moves, until turn > MAX_TURNS) of
    1600-rollout searches with the tree re-used from move to move, recorded from the reference itself
    (tools/gen_golden.py --only-game -> tests/golden/mcts_trace_game.json; 3,687 value + 845 policy evaluations there).
    Every move, every root-child visit count, the root's visit count and winrate before each move, and the final board are
    the reference's, on the Python tree and on the native tree (its default: evaluation ahead of expansion, requests held
    to the cooperative launch's size steps)."""
    from bokego_amd import nnet
    from bokego_amd.mcts import MCTS, Go_MCTS
    from bokego_amd.mcts_native import NativeMCTS, Position
    t = json.load(open(os.path.join(GOLDEN, "mcts_trace_game.json")))["r1600_game"]
    assert len(t["moves"]) == 81 and t["rollouts"] == 1600
    pi, val = nnet.HipPolicyNet(sds[0]), nnet.HipValueNet(sds[1])
    torch.manual_seed(0)
    ms_py = _replay_whole_game(MCTS(Go_MCTS(), pi, val, no_sim=True), t, native=False) * 1e3
    nat = NativeMCTS(Position(), pi, val)
    ms_nat = _replay_whole_game(nat, t, native=True) * 1e3
    gi = nat._pool.info(0)
    print(f"\nconfig3 whole game (81 moves): Python tree {ms_py:.1f} ms/move, native tree {ms_nat:.2f} ms/move "
          f"({gi['n_requests'] / 81:.1f} requests, {gi['n_value_evals'] / 81:.0f} value rows per move); "
          f"reference: {t['n_value_evals']} value / {t['n_policy_evals']} policy evaluations, ~0.9 s/move")
    out = os.path.join(os.path.dirname(os.path.dirname(GOLDEN)), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    if os.path.isdir(out):
        with open(os.path.join(out, "cfg2_whole_game.json"), "w") as f:
            json.dump({"moves": 81, "rollouts": 1600, "python_tree_ms_per_move": ms_py, "native_tree_ms_per_move": ms_nat,
                       "native_requests_per_move": gi["n_requests"] / 81, "native_value_rows_per_move": gi["n_value_evals"] / 81,
                       "identical_to_reference": True}, f)


def test_native_tree_snapshot_continues_the_reference_trace(sds):
    """NativeMCTS -- the tree the default launcher runs -- pickles and deep-copies like the reference's MCTS
    (mcts.py:81-108: the nets do not travel; a deep copy shares them).  BASELINE config 3's recorded trace is played to move 5
    on the HIP nets, the tree pickled, unpickled, given its nets back: moves 5..9 and every root-child visit count are still
    the reference's; so are they on a deep copy taken at the same point, and on the original afterwards."""
    import copy
    import pickle
    from bokego_amd import nnet
    from bokego_amd.mcts_native import NativeMCTS, Position
    t = json.load(open(os.path.join(GOLDEN, "mcts_trace.json")))["r1600"]
    pi, val = nnet.HipPolicyNet(sds[0]), nnet.HipValueNet(sds[1])

    def step(tree, ref):
        tree.rollout(t["rollouts"])
        assert {m: n for m, (n, _) in tree.child_stats().items()} == {int(k): v for k, v in ref["child_N"].items()}
        assert abs(tree.winrate() - ref["root_winrate"]) < 1e-4
        assert tree.choose().last_move == ref["move"]

    tree = NativeMCTS(Position(), pi, val)
    for ref in t["moves"][:5]:
        step(tree, ref)
    blob = pickle.dumps(tree)
    twin = copy.deepcopy(tree)
    back = pickle.loads(blob)
    assert back.policy_net is None and back.evaluator is None and len(blob) > 100_000
    back.policy_net, back.value_net = pi, val
    for who in (back, twin, tree):
        for ref in t["moves"][5:]:
            step(who, ref)
    assert back.evaluator.engine is tree.evaluator.engine          # rebuilt from the same fused nets
    assert back.child_stats() == twin.child_stats() == tree.child_stats()


@pytest.mark.parametrize("precision,kw", [("f16x2", {}), ("f32", {"speculate": 60, "speculate_rows": 256, "request_tasks": 0}), ("f32", {})])
def test_config3_with_evaluation_ahead_of_expansion(sds, precision, kw):
    """search_params.speculate (the default of NativeMCTS on an f16x2 engine; forced here for fp32 in its whole-candidate
    form; and the fp32 default: candidates staged within the request-size steps 64 / 80 / 128): likely-to-be-expanded leaves
    are evaluated with requests that go out anyway.  BASELINE config 3 still reproduces the reference trace -- every move,
    every root-child visit count -- with fewer requests than the plain search."""
    from bokego_amd import nnet
    from bokego_amd.mcts_native import NativeMCTS, Position
    t = json.load(open(os.path.join(GOLDEN, "mcts_trace.json")))["r1600"]
    mk = lambda **k: NativeMCTS(Position(), nnet.HipPolicyNet(sds[0], precision=precision),  # noqa: E731
                                nnet.HipValueNet(sds[1], precision=precision), **k)
    ahead, plain = mk(**kw), mk(speculate=0)
    for ref in t["moves"]:
        for tree in (ahead, plain):
            tree.rollout(t["rollouts"])
            kids = {m: n for m, (n, _) in tree.child_stats().items()}
            assert kids == {int(k): v for k, v in ref["child_N"].items()}
            assert tree.choose().last_move == ref["move"]
    a, p = ahead._pool.info(0), plain._pool.info(0)
    if kw or precision == "f16x2":
        assert a["n_requests"] < 0.8 * p["n_requests"] and a["n_value_evals"] > p["n_value_evals"]
    else:       # in the opening's 80-child positions a 64-task request has little room for passengers: fewer, not far fewer
        assert a["n_requests"] <= p["n_requests"] and a["n_value_evals"] >= p["n_value_evals"]


def test_analyze_and_tree_views_on_the_native_tree_on_gpu(sds):
    """VERDICT r2 item 4 on the HIP engine: the default launcher's tree (NativeGTP) serves `analyze` from the search it
    is running (gtp.py:374-399) -- the same info line as the Python tree prints after the same rollouts -- and lets
    callers read N / V / children of any node (gtp.py:386,395 read self.N[n])."""
    from bokego_amd import nnet
    from bokego_amd.gtp import GTP, NativeGTP
    from bokego_amd.mcts import Go_MCTS
    from bokego_amd.mcts_native import Position
    mk = lambda cls, root: cls(root, nnet.HipPolicyNet(sds[0]), nnet.HipValueNet(sds[1]), no_sim=True, time_lim=None, n_rollouts=200)  # noqa: E731
    py, nat = mk(GTP, Go_MCTS()), mk(NativeGTP, Position())
    out = []
    for g in (py, nat):
        g.running = True
        assert g.send("play b e5") == "= \n\n"
        var = {}
        g.rollout(1600, analyze_dict=var)
        out.append((g.analyze_line(var, k=3), {n.last_move: [m.last_move for m in line] for n, line in var.items()}))
    assert out[0] == out[1] and out[1][0].count("info move") == 3
    for node, n in py.N.items():
        assert nat.N[node] == n and abs(nat.V[node] - py.V[node]) < 1e-9
    assert {k.key() for k in nat.children[nat.root]} == {k.key() for k in py.children[py.root]}
    assert torch.equal(nat.root.dist.probs, py.root.dist.probs)
    gen = nat.send("analyze w 10")
    assert next(gen) == "= \n" and next(gen).startswith("info move ")


def test_branch_num_and_choose_below_the_root_on_the_engine(sds):
    """The two kwargs / calls the native tree gained in round 4, on the product path (HIP nets, GPU-encoded planes): branch_num = 8
    (mcts.py:62,189-190) gives the Python tree's search visit for visit over four 600-rollout moves, and choose(node) for a child
    of the root (mcts.py:110-131) returns the same grandchild on both trees without moving either root."""
    from bokego_amd import nnet
    from bokego_amd.mcts import MCTS, Go_MCTS
    from bokego_amd.mcts_native import NativeMCTS, Position
    pi, val = nnet.HipPolicyNet(sds[0]), nnet.HipValueNet(sds[1])
    py = MCTS(Go_MCTS(), pi, val, branch_num=8)
    nat = NativeMCTS(Position(), pi, val, branch_num=8)
    for ply in range(4):
        py.rollout(600); nat.rollout(600)
        want = {c.mv: (py.N[c], py.V[c]) for c in py.children[py.root]}
        assert nat.child_stats() == want and 0 < len(want) <= 8, ply
        root_key = nat.root.key()
        for c in py.children[py.root]:
            if c in py.children and py.children[c]:
                assert nat.choose(nat.root.make_move(c.last_move)).key() == py.choose(c).key()
        assert nat.root.key() == root_key == py.root.key()
        assert py.choose().last_move == nat.choose().last_move


def test_simulation_mode_on_the_native_tree_on_the_engine(sds):
    """boke.py --simulate (MCTS(no_sim=False), mcts.py:147-148,195-217) on the native tree with the HIP engine behind it: every
    rollout ends in a policy playout evaluated position by position on the GPU.  (a) both nets, value_net_weight 0.5: N / Q /
    V of the root's children equal the sequential restatement (oracle/mcts_ref.py) fed by the CPU oracle nets with the same
    generator -- a draw would have to land within ~1e-6 of a boundary of the cumulative distribution to differ; (b) the policy
    net alone (an engine without a ValueNet): Q only, V stays 0; (c) the same seed gives the same search, another seed another."""
    from bokego_amd import nnet
    from bokego_amd.mcts_native import NativeMCTS, Position
    from oracle.mcts_ref import RefMCTS
    from oracle.oracle import OraclePolicy, OracleValue
    P, V = OraclePolicy(sds[0]), OracleValue(sds[1])
    pi, val = nnet.HipPolicyNet(sds[0]), nnet.HipValueNet(sds[1])
    ref = RefMCTS(P, V, expand_thresh=3, simulate=True, seed=21)
    nat = NativeMCTS(Position(), pi, val, no_sim=False, expand_thresh=3, seed=21)
    t0 = time.perf_counter()
    nat.rollout(16)
    dt = time.perf_counter() - t0
    ref.rollout(16)
    assert nat.value_net_weight == 0.5 and nat.N[nat.root] == 16
    for mv, ck in ref.children[ref.root]:
        child = nat.root.make_move(mv)
        assert nat.N[child] == ref.N.get(ck, 0) and nat.Q[child] == ref.Q.get(ck, 0.0), mv
        assert abs(nat.V[child] - ref.V.get(ck, 0.0)) < 1e-3
    assert nat.choose().last_move == ref.choose()
    gi = nat._pool.info(0)
    print(f"simulation mode: 16 rollouts in {dt * 1e3:.0f} ms, {gi['n_policy_evals']} policy rows in {gi['n_requests']} requests")

    only = [NativeMCTS(Position(), nnet.HipPolicyNet(sds[0]), None, no_sim=False, expand_thresh=3, seed=s) for s in (5, 5, 6)]
    stats = []
    for t in only:
        t.rollout(24)
        kids = [t.root.make_move(mv) for mv in t.child_stats()]
        assert t.value_net_weight == 0.0 and t.root.value is None
        assert sum(t.N[c] for c in kids) == 24 == t.N[t.root] and all(t.V[c] == 0 for c in kids)
        assert sum(t.Q[c] for c in kids) == -t.Q[t.root] and abs(t.Q[t.root]) <= 24
        stats.append({c.last_move: (t.N[c], t.Q[c]) for c in kids})
    assert stats[0] == stats[1] and stats[0] != stats[2]
