"""-m gpu: the callers of the path on the HIP nets -- BASELINE configs[3] at its stated size (512 games x 400
rollouts/move), the reference's policy playouts (bin/selfplay.py:18-57) seed-matched against oracle-backed nets,
and the on-disk formats end to end (.pt checkpoint -> launcher -> engine; self-play records re-read)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from bokego_amd import go, nnet, selfplay
from bokego_amd.bkw import load_bkw, tensors_to_state_dict

from conftest import GOLDEN, REPO

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sds():
    return load_bkw(os.path.join(GOLDEN, "policy_19.bkw")), load_bkw(os.path.join(GOLDEN, "value_synth.bkw"))


@pytest.fixture(scope="module")
def engine(sds):
    from bokego_amd.engine import LeafEngine
    e = LeafEngine(sds[0], sds[1], max_batch=8192)      # default precision: fp32, the reference's width
    assert e.precision == "f32"
    yield e
    e.close()


# ---- configs[3] at full size ------------------------------------------------------------------------------------------
CFG4 = dict(n_games=512, rollouts=400, cap=8192)


@pytest.fixture(scope="module")
def generation(engine):
    """ONE rank playing all 512 games (what bench.py's selfplay block times)."""
    local, total = selfplay.self_play(selfplay.EngineEvaluator(engine), record_visits=1, **CFG4)
    return local, total


def test_config4_full_size_is_independent_of_sharding(engine, generation):
    """512 games x 400 rollouts/move: the same games played as the two shards of a world-size-2 job, one after the
    other on this GPU, give the same per-game move lists, and their statistics add up to the one-rank generation's
    (SURVEY 8d config 4's invariant: `gid % world`, seeds `seed_base + gid`, selfplay.py:177-199)."""
    local, total = generation
    assert total["games"] == 512 and len(local["games"]) == 512
    assert 512 * 40 < total["plies"] <= 512 * 81 and total["black_wins"] + total["white_wins"] == 512
    assert sum(total["first_move_hist"]) == 512
    # visit / value statistics in the reduced vector (north_star; VERDICT r4 missing #3): one root value per ply, ~400 rollouts
    # per ply among the root's children (the root's own visits carried over from the previous move's subtree come on top)
    assert total["n_root_values"] == total["plies"] and 0 < total["mean_abs_root_value"] < 1
    assert 380 * total["plies"] < sum(total["root_visit_hist"]) < 2000 * total["plies"]
    hist = np.zeros(81, np.int64)
    for plies in local["visits"].values():
        for v in plies:
            for mv, n in v.items():
                hist[mv] += n
    assert total["root_visit_hist"] == hist.tolist()
    assert len({tuple(g["moves"]) for g in local["games"].values()}) > 400        # the games really differ
    shards, stats = {}, np.zeros(selfplay.STATS_LEN)
    for rank in range(2):
        loc, _ = selfplay.self_play(selfplay.EngineEvaluator(engine), rank=rank, world=2, **CFG4)
        assert sorted(loc["games"]) == selfplay.shard_game_ids(512, rank, 2)
        shards.update(loc["games"])
        stats += loc["local_stats"]
    assert shards == local["games"]                                                # (i) move lists and scores, every game
    assert np.array_equal(stats, local["local_stats"])                             # (ii) what the all-reduce would sum
    named = {k: float(stats[i]) for i, k in enumerate(selfplay.STATS_FIELDS)}
    assert all(named[k] == total[k] for k in selfplay.STATS_FIELDS)


def test_config4_does_not_depend_on_what_is_evaluated_when(engine, generation):
    """Round 3: an expansion evaluates its 4 best-prior children (bk_search_params.eager_top), the rest when the search gets
    there, and fp32 batches stop at whole rounds of workgroups (bk_pool_set_task_cap) -- the defaults the `generation` fixture
    ran with.  The same 512 x 400 generation with EVERY child evaluated at its parent's expansion and no cap (rounds 1-2), and
    with other settings in between, plays the same games move for move with the same root visit counts; only the number of
    evaluations differs (3.4 M -> 0.7 M)."""
    local, total = generation
    assert local["native_loop"]                      # the fixture ran the step loop in C (bk_pools_run + the engine's bk_evaluator)
    for kw in (dict(eager_top=0, task_cap=0), dict(eager_top=8, task_cap=0, n_pools=3), dict(eager_top=2, task_cap=500),
               dict(native_loop=False), dict(native_loop=False, n_pools=3, eager_top=4)):     # ... and the same loop in Python
        loc, tot = selfplay.self_play(selfplay.EngineEvaluator(engine), record_visits=1, **CFG4, **kw)
        assert loc["games"] == local["games"] and loc["visits"] == local["visits"], kw
        assert all(tot[k] == total[k] for k in ("games", "black_wins", "white_wins", "plies", "sum_score", "first_move_hist",
                                                 "root_visit_hist", "sum_root_value", "sum_abs_root_value", "n_root_values")), kw
        if kw.get("eager_top", 2) == 2 and "task_cap" not in kw:
            assert tot == total, kw                   # same evaluation schedule: every counter of the reduced vector
        if kw.get("eager_top") == 0:
            assert tot["value_evals"] > 4 * total["value_evals"]


ORACLE_GIDS = list(range(5, 512, 16))          # 32 of the 512 games, spread over the shards of every world size


def test_config4_whole_games_equal_the_cpu_oracle_run(generation, sds):
    """(iii) anchored to the oracle, not to the HIP path itself (VERDICT r4 next #1a): the native pool driven by the CPU
    oracle nets (oracle/nnet_ref.c through CallableEvaluator) plays 32 of the 512 games of the SAME generation TO THE END
    -- same seeds, 400 rollouts/move, Dirichlet noise and visit sampling included -- and every move and every root-child
    visit count of every ply must equal the GPU run's.  The two sides' network outputs differ by ~3e-5, so a PUCT near-tie
    may resolve the other way round: at least 30 of the 32 games must be identical, and where one is not, its first
    divergent ply must be a visit tie within 1 (one rollout gone to a sibling).  The figures go to
    gpurun_out/cfg3_oracle_depth.json (committed under profiles/)."""
    import time
    from oracle.oracle import OraclePolicy, OracleValue, set_threads
    set_threads(min(16, len(os.sched_getaffinity(0))))
    P, V = OraclePolicy(sds[0]), OracleValue(sds[1])
    local, _ = generation
    prm = selfplay.search_params(rollouts=400, expand_thresh=100, noise_weight=0.25, sample_plies=8,
                                 max_turns=80, prune=1, record_visits=1, eager_top=2)     # self_play()'s parameters
    pool = selfplay.GamePool([20260 + g for g in ORACLE_GIDS], prm, cap=8192)
    ev = selfplay.CallableEvaluator(P, V)
    t0 = time.perf_counter()
    selfplay.run_pools([pool], ev)
    secs = time.perf_counter() - t0
    identical, plies, diverged = 0, 0, []
    for i, g in enumerate(ORACLE_GIDS):
        want_m, want_v = local["games"][g]["moves"], local["visits"][g]
        got_m = pool.moves(i)
        got_v = [pool.visits(i, ply) for ply in range(len(got_m))]
        first = next((k for k in range(max(len(got_m), len(want_m)))
                      if k >= len(got_m) or k >= len(want_m) or got_m[k] != want_m[k] or got_v[k] != want_v[k]), None)
        if first is None:
            identical += 1
            plies += len(got_m)
            assert pool.info(i)["score"] == local["games"][g]["score"]
            continue
        plies += first
        a, b = (got_v[first] if first < len(got_v) else {}), (want_v[first] if first < len(want_v) else {})
        dn = {m: a.get(m, 0) - b.get(m, 0) for m in set(a) | set(b) if a.get(m, 0) != b.get(m, 0)}
        diverged.append({"gid": g, "ply": first, "delta_N": {str(k): v for k, v in dn.items()},
                         "moves": [got_m[first] if first < len(got_m) else None, want_m[first] if first < len(want_m) else None]})
    rec = {"games": len(ORACLE_GIDS), "identical_games": identical, "oracle_anchored_plies": plies, "diverged": diverged,
           "oracle_seconds": secs, "oracle_positions": ev.positions, "oracle_threads": min(16, len(os.sched_getaffinity(0)))}
    out = os.path.join(REPO, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "cfg3_oracle_depth.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print("\ncfg[3] oracle depth:", {k: v for k, v in rec.items() if k != "diverged"}, diverged)
    pool.close()
    assert identical >= 30 and plies >= 2000, rec
    for d in diverged:
        assert d["delta_N"] and max(abs(v) for v in d["delta_N"].values()) <= 1 and len(d["delta_N"]) <= 2, d


def test_config4_f16x2_plays_the_same_generation(engine, generation):
    """The opt-in f16x2 arithmetic: same engine, same seeds -> the same 512 games move for move."""
    local, total = generation
    engine.set_precision("f16x2")
    try:
        loc, tot = selfplay.self_play(selfplay.EngineEvaluator(engine), **CFG4)
    finally:
        engine.set_precision("f32")
    same = sum(loc["games"][g]["moves"] == local["games"][g]["moves"] for g in range(512))
    assert same == 512, same
    assert tot["plies"] == total["plies"] and tot["black_wins"] == total["black_wins"]
    assert engine.stats()["f16_overflow_fallbacks"] == 0


def test_selfplay_records_reread_and_replayed(generation, tmp_path):
    """f4: records written by a GPU generation (one SGF per game + games.json with per-ply root visit counts) are
    read back with the SGF reader and replayed legally on a fresh board, to the same final score."""
    local, _ = generation
    gids = list(range(0, 512, 37))
    selfplay.write_records(str(tmp_path), {g: local["games"][g] for g in gids}, {g: local["visits"][g] for g in gids})
    rec = json.load(open(tmp_path / "games.json"))
    assert sorted(map(int, rec)) == gids
    for g in gids:
        moves = rec[str(g)]["moves"]
        assert moves == local["games"][g]["moves"]
        assert go.get_moves(str(tmp_path / f"game_{g:05d}.sgf")) == moves
        board = go.Game(moves=[])
        for m in moves:
            assert m == go.PASS or board.is_legal(m)
            board.play_move(m)
        assert board.area_score() == rec[str(g)]["score"]
        vis = rec[str(g)]["visits"]
        assert len(vis) == len(moves)
        for ply, m in enumerate(moves[:-1]):
            if m != go.PASS and vis[ply]:
                v = {int(k): n for k, n in vis[ply].items()}
                assert m in v and sum(v.values()) >= 300        # 400 rollouts/move, minus the root's own visits


# ---- a6 / a11: the reference's policy playouts, seed-matched ------------------------------------------------------------
class _OracleNet:
    """policy_net(x[B,27,9,9]) -> logits, backed by the CPU oracle (the reference's arithmetic)."""

    def __init__(self, sd):
        from oracle.oracle import OraclePolicy
        self.fn = OraclePolicy(sd)

    def to(self, d):
        return self

    def eval(self):
        return self

    def __call__(self, x):
        return torch.from_numpy(self.fn(x.detach().cpu().numpy()))


def test_policy_sample_legal_sample_playout_match_oracle_backed_net(sds):
    """nnet.policy_sample (a6, nnet.py:286-297), selfplay.legal_sample / playout / policy_self_play (a11,
    bin/selfplay.py:18-57): with the same torch seed the HIP net and an oracle-backed net draw the same moves."""
    hip, ora = nnet.HipPolicyNet(sds[0]), _OracleNet(sds[0])
    f = np.load(os.path.join(GOLDEN, "features.npz"))["incremental"].astype(np.float32)
    for i in (0, 40, 200, 333, 500):
        fts = torch.from_numpy(f[i])
        for seed in (1, 2, 3):
            torch.manual_seed(seed)
            a = nnet.policy_sample(hip, None, fts=fts)
            torch.manual_seed(seed)
            b = nnet.policy_sample(ora, None, fts=fts)
            assert a.dtype == torch.int64 and a.shape == () and int(a) == int(b)
    game = go.Game(moves=[])
    for seed in range(4):
        torch.manual_seed(seed)
        a = selfplay.legal_sample(hip, game)
        torch.manual_seed(seed)
        b = selfplay.legal_sample(ora, game)
        assert int(a) == int(b) and game.is_legal(int(a))
    # a position with few legal moves left: the descending-probability walk (selfplay.py:40-46) must agree too
    res = {}
    for name, net in (("hip", hip), ("oracle", ora)):
        torch.manual_seed(11)
        res[name] = selfplay.policy_self_play(net, net, 3)
    assert res["hip"] == res["oracle"]
    games, results = res["hip"]
    assert len(games) == 3 and all(r in (1, -1) for r in results)
    for mv in games:
        assert 60 <= len(mv) <= 71                              # turn > 70 ends a playout (selfplay.py:16,22)
        g = go.Game(moves=[])
        for m in mv:
            assert g.is_legal(m)
            g.play_move(m)
    hip.engine().close()


# ---- f4: a reference-style .pt checkpoint through the launcher -----------------------------------------------------------
def test_pt_checkpoint_through_launcher_to_engine(sds, tmp_path):
    """boke.py:30-38's contract: torch.load(path)["model_state_dict"] -> load_state_dict -> eval.  Checkpoints written
    in the reference's format load through the launcher's loader into the HIP nets (empty-board logits vs the
    goldens), and `python -m bokego_amd.gtp -p x.pt -v y.pt -r N` answers a GTP session with them."""
    from bokego_amd.gtp import load_state_dict
    ppt, vpt = str(tmp_path / "policy.pt"), str(tmp_path / "value.pt")
    torch.save({"model_state_dict": tensors_to_state_dict(sds[0]), "optimizer_state_dict": {}, "epoch": 19}, ppt)
    torch.save({"model_state_dict": tensors_to_state_dict(sds[1])}, vpt)
    pi = nnet.HipPolicyNet()
    pi.load_state_dict(load_state_dict(ppt))
    pi.eval()
    v = nnet.HipValueNet()
    v.load_state_dict(load_state_dict(vpt))
    n = np.load(os.path.join(GOLDEN, "nets.npz"))
    fts = nnet.features(go.Game())
    assert fts.sum() == 531
    lg = pi(fts.unsqueeze(0))
    assert np.abs(lg.numpy()[0] - n["logits_b1"][0]).max() < 1e-4 and abs(lg[0, 40].item() - 12.709958) < 1e-4
    assert abs(nnet.value(v, go.Game()) - float(n["values_b1"][0])) < 1e-4
    d = nnet.policy_dist(pi, go.Game())
    assert abs(d.probs[40].item() - 0.818645) < 1e-5
    pi.engine().close(); v.engine().close()
    session = "name\nkomi 7.5\nplay b e5\ngenmove w\nfinal_score\nmove_history\nquit\n"
    for extra in ([], ["--python-tree"]):
        out = subprocess.run([sys.executable, "-m", "bokego_amd.gtp", "-p", ppt, "-v", vpt, "-r", "200"] + extra,
                             input=session, capture_output=True, text=True, timeout=300, cwd=REPO)
        assert out.returncode == 0, out.stderr[-2000:]
        rep = out.stdout.split("\n\n")
        assert rep[0] == "= boke" and rep[1] == "= " and rep[2] == "= "
        assert rep[3] == "= G5"                                  # the reference's reply to E5 at 200 rollouts (gtp_transcript.json)
        assert rep[4].startswith("= W+") or rep[4].startswith("= B+")
        assert rep[5] == "= E5\nG5"


def test_worker_team_on_the_gpu_host():
    """tests/test_selfplay_cpu.py::test_worker_team_survives_late_workers again, on the GPU box's own CPU: the hand-shake
    bug it guards against (a worker walking into a retired parallel region) showed within a few hundred regions on the
    EPYC hosts of this pool and never on the build container's Xeon.  Needs no GPU; marked gpu to run where it matters."""
    lib = selfplay.treelib()
    for threads in (2, 3, 6, 12):
        assert lib.bk_team_selftest(threads, 300_000) == 0


def test_in_batch_deduplication_on_the_engine_plays_the_same_games():
    """bk_pool_set_dedup on the real path (fp32 engine, planes encoded on the GPU from the records that travel): 128 games x 400
    rollouts/move with and without it -- the same games move for move, the same per-game evaluation counts, fewer rows sent; and
    the default follows the engine's precision (on for fp32, off for f16x2)."""
    from bokego_amd.engine import LeafEngine
    eng = LeafEngine(load_bkw(os.path.join(GOLDEN, "policy_19.bkw")), load_bkw(os.path.join(GOLDEN, "value_synth.bkw")), max_batch=8192)
    ev = selfplay.EngineEvaluator(eng)
    runs = {}
    for name, dd in (("plain", False), ("dedup", True), ("default", None)):
        local, total = selfplay.self_play(ev, n_games=128, rollouts=400, cap=8192, dedup=dd)
        runs[name] = (local, total)
    base = runs["plain"]
    for name in ("dedup", "default"):
        local, total = runs[name]
        assert local["games"] == base[0]["games"] and total == base[1]
        assert local["dedup"] and 0 < local["rows_sent"] < local["rows_requested"]
    assert not base[0]["dedup"] and base[0]["rows_requested"] == 0
    eng.set_precision("f16x2")
    local, _ = selfplay.self_play(selfplay.EngineEvaluator(eng), n_games=16, rollouts=50, cap=8192)
    assert not local["dedup"]
    eng.close()


def test_a_ranks_share_at_eight_ranks_plays_its_games_with_evaluation_ahead(engine, generation):
    """configs[3] at 8 ranks: rank 3's 64 games (two pools of 32: ~94-task requests, the 2-CUs-per-board launch) run with
    evaluation ahead of expansion and batches held to 128 tasks by default (selfplay.small_shard_defaults; 0.181 -> 0.171 s,
    profiles/r05_spec_probe.txt).  They are the games the one-rank generation played for those ids, move for move and visit for
    visit, and the same as with evaluation ahead switched off -- in fewer steps."""
    local, _ = generation
    kw = dict(rank=3, world=8, record_visits=1, **CFG4)
    on, _ = selfplay.self_play(selfplay.EngineEvaluator(engine), **kw)
    off, _ = selfplay.self_play(selfplay.EngineEvaluator(engine), speculate=0, **kw)
    assert on["speculate"] == 70 and on["task_caps"] == [128, 128] and off["speculate"] == 0
    gids = selfplay.shard_game_ids(512, 3, 8)
    assert sorted(on["games"]) == gids and on["games"] == off["games"] == {g: local["games"][g] for g in gids}
    assert on["visits"] == off["visits"] == {g: local["visits"][g] for g in gids}
    assert on["steps"] < off["steps"]


def test_the_opt_in_multi_leaf_mode_on_the_engine(engine, generation):
    """bk_search_params.leaves (virtual loss only as an opt-in throughput mode; outside SURVEY 8, no parity claim): a rank's
    64-game share of configs[3] with 8 leaves per tree and step on the real path.  Still a pure function of the seeds -- the same games
    for another pool split, the step loop in Python, and as two half-shards -- with fewer, larger requests than the one-leaf search,
    every ply's record holding that ply's 400 rollouts; and the default (leaves = 1) is untouched: the `generation` fixture's games."""
    kw = dict(n_games=512, rollouts=400, cap=8192, rank=0, world=8, record_visits=1)
    one, _ = selfplay.self_play(selfplay.EngineEvaluator(engine), **kw)
    assert one["leaves"] == 1 and all(one["games"][g] == generation[0]["games"][g] for g in one["games"])
    base, tot = selfplay.self_play(selfplay.EngineEvaluator(engine), leaves=8, **kw)
    assert base["leaves"] == 8 and base["games"] != one["games"]
    assert base["steps"] < 0.75 * one["steps"] and base["rows_sent"] / base["steps"] > 1.2 * one["rows_sent"] / one["steps"]
    for extra in (dict(n_pools=3), dict(native_loop=False, threads=2)):
        loc, t2 = selfplay.self_play(selfplay.EngineEvaluator(engine), leaves=8, **kw, **extra)
        assert loc["games"] == base["games"] and loc["visits"] == base["visits"] and t2["root_visit_hist"] == tot["root_visit_hist"], extra
    halves = {}
    for r in (0, 8):                                   # ranks 0 and 8 of a world of 16 = the two halves of rank 0's share of 8
        loc, _ = selfplay.self_play(selfplay.EngineEvaluator(engine), leaves=8, n_games=512, rollouts=400, cap=8192, rank=r, world=16)
        halves.update(loc["games"])
    assert halves == base["games"]
    assert all(sum(ply.values()) >= 399 for plies in base["visits"].values() for ply in plies)
