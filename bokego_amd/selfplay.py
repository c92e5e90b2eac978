"""Self-play with MCTS on many concurrent games, sharded over GPUs (BASELINE config 4).

The reference's bin/selfplay.py plays raw policy-vs-policy games in `cpu_count()` processes with
shared CPU tensors (selfplay.py:18-57,177-199) and never uses MCTS; config 4 composes its game
loop with `MCTS`.  Here:

  * every rank (one process per GPU) owns the games `gid % world == rank`; a game's randomness
    comes only from `seed_base + gid`, so the set of games does not depend on the world size;
  * a rank's games live in native lock-step pools (include/bokego_tree.h): each step gathers the
    pending leaf evaluations of ALL its games into one batch for the HIP engine; two pools
    alternate so the host advances one while the GPU evaluates the other;
  * there is no data-path collective.  At the end of the generation ONE all-reduce (RCCL over
    xGMI on GPUs, gloo in the CPU tests) sums a small statistics vector.
"""
import argparse
import ctypes
import json
import os
import time

import numpy as np

from . import go

# ---- ctypes view of include/bokego_tree.h ------------------------------------------------------------


class SearchParams(ctypes.Structure):
    _fields_ = [("rollouts", ctypes.c_int32), ("expand_thresh", ctypes.c_int32), ("c_puct", ctypes.c_double),
                ("noise_weight", ctypes.c_float), ("sample_plies", ctypes.c_int32), ("max_turns", ctypes.c_int32),
                ("eager", ctypes.c_int32), ("komi", ctypes.c_float), ("record_visits", ctypes.c_int32),
                ("prune", ctypes.c_int32), ("speculate", ctypes.c_int32),
                ("speculate_rows", ctypes.c_int32), ("request_tasks", ctypes.c_int32), ("eager_top", ctypes.c_int32),
                ("request_steps", ctypes.c_int32 * 3), ("branch_num", ctypes.c_int32),
                ("simulate", ctypes.c_int32), ("use_value", ctypes.c_int32), ("value_weight", ctypes.c_double),
                ("leaves", ctypes.c_int32), ("leaves_visit_only", ctypes.c_int32)]


class NodeInfo(ctypes.Structure):
    _fields_ = [("N", ctypes.c_int32), ("n_children", ctypes.c_int32), ("V", ctypes.c_double), ("value", ctypes.c_float),
                ("move", ctypes.c_int16), ("flags", ctypes.c_uint16)]


class GameInfo(ctypes.Structure):
    _fields_ = [("done", ctypes.c_int32), ("n_moves", ctypes.c_int32), ("score", ctypes.c_float),
                ("n_nodes", ctypes.c_int32), ("n_value_evals", ctypes.c_uint64), ("n_policy_evals", ctypes.c_uint64),
                ("n_requests", ctypes.c_uint64), ("root_N", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("root_V", ctypes.c_double)]


class GameStats(ctypes.Structure):
    _fields_ = [("root_visits", ctypes.c_uint64 * 81), ("sum_root_value", ctypes.c_double),
                ("sum_abs_root_value", ctypes.c_double), ("n_root_values", ctypes.c_uint64)]


_SUBMIT_FN = ctypes.CFUNCTYPE(ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)
_WAIT_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int64)


class EvaluatorStruct(ctypes.Structure):
    """bk_evaluator (include/bokego_tree.h): what bk_pools_run calls to have a batch evaluated"""
    _fields_ = [("ctx", ctypes.c_void_p), ("submit", _SUBMIT_FN), ("wait", _WAIT_FN)]


class RunInfo(ctypes.Structure):
    _fields_ = [("steps", ctypes.c_uint64), ("rows", ctypes.c_uint64), ("policy_rows", ctypes.c_uint64),
                ("seconds", ctypes.c_double), ("wait_seconds", ctypes.c_double)]


_VP = ctypes.c_void_p
TREE_SYMBOLS = {
    "bk_search_params_default": (None, [ctypes.POINTER(SearchParams)]),
    "bk_pool_create": (_VP, [ctypes.c_int, ctypes.POINTER(SearchParams), _VP, ctypes.c_int]),
    "bk_pool_destroy": (None, [_VP]),
    "bk_pool_collect": (ctypes.c_int, [_VP, _VP, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    "bk_pool_collect_pos": (ctypes.c_int, [_VP, _VP, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    "bk_pool_deliver": (None, [_VP, _VP, _VP]),
    "bk_pool_phase_seconds": (None, [_VP, _VP]),
    "bk_pool_set_dedup": (None, [_VP, ctypes.c_int]),
    "bk_pool_set_lanes": (None, [_VP, ctypes.c_int]),
    "bk_pool_dedup_rows": (None, [_VP, _VP, _VP]),
    "bk_team_selftest": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "bk_team_selftest_concurrent": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "bk_pool_set_task_cap": (None, [_VP, ctypes.c_int]),
    "bk_pool_n_games": (ctypes.c_int, [_VP]),
    "bk_pool_n_done": (ctypes.c_int, [_VP]),
    "bk_pool_game_info": (ctypes.c_int, [_VP, ctypes.c_int, ctypes.POINTER(GameInfo)]),
    "bk_pool_game_moves": (ctypes.c_int, [_VP, ctypes.c_int, _VP, ctypes.c_int]),
    "bk_pool_root_children": (ctypes.c_int, [_VP, ctypes.c_int, _VP, _VP, _VP]),
    "bk_pool_game_visits": (ctypes.c_int, [_VP, ctypes.c_int, ctypes.c_int, _VP, _VP]),
    "bk_pool_set_manual": (None, [_VP, ctypes.c_int]),
    "bk_pool_add_rollouts": (ctypes.c_int, [_VP, ctypes.c_int, ctypes.c_int]),
    "bk_pool_choose": (ctypes.c_int, [_VP, ctypes.c_int]),
    "bk_pool_play": (ctypes.c_int, [_VP, ctypes.c_int, ctypes.c_int]),
    "bk_pool_set_position": (ctypes.c_int, [_VP, ctypes.c_int, _VP]),
    "bk_pool_root_pos": (ctypes.c_int, [_VP, ctypes.c_int, _VP]),
    "bk_pool_set_speculation": (None, [_VP, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "bk_pool_find": (ctypes.c_int, [_VP, ctypes.c_int, _VP]),
    "bk_pool_root_id": (ctypes.c_int, [_VP, ctypes.c_int]),
    "bk_pool_node": (ctypes.c_int, [_VP, ctypes.c_int, ctypes.c_int, ctypes.POINTER(NodeInfo), _VP]),
    "bk_pool_node_children": (ctypes.c_int, [_VP, ctypes.c_int, ctypes.c_int, _VP, ctypes.c_int]),
    "bk_pool_node_prior": (ctypes.c_int, [_VP, ctypes.c_int, ctypes.c_int, _VP]),
    "bk_pool_node_q": (ctypes.c_int, [_VP, ctypes.c_int, ctypes.c_int, _VP]),
    "bk_pool_principal_variation": (ctypes.c_int, [_VP, ctypes.c_int, _VP, ctypes.c_int]),
    "bk_pool_set_analyze": (None, [_VP, ctypes.c_int]),
    "bk_pool_variation": (ctypes.c_int, [_VP, ctypes.c_int, ctypes.c_int, _VP, ctypes.c_int]),
    "bk_pool_game_stats": (ctypes.c_int, [_VP, ctypes.c_int, ctypes.POINTER(GameStats)]),
    "bk_pool_snapshot": (ctypes.c_long, [_VP, ctypes.c_int, _VP, ctypes.c_long]),
    "bk_pool_restore": (ctypes.c_int, [_VP, ctypes.c_int, _VP, ctypes.c_long]),
    # (the evaluator as a plain pointer: under `python -m bokego_amd.selfplay` this module exists twice -- as __main__ and as
    # bokego_amd.selfplay, which engine.evaluator() imports -- and a typed pointer would reject the other copy's struct)
    "bk_pools_run": (ctypes.c_int, [_VP, ctypes.c_int, _VP, ctypes.c_int, ctypes.POINTER(RunInfo)]),
    "bk_normalise_rows": (None, [_VP, ctypes.c_int]),
}
_tree_ready = False


def treelib():
    global _tree_ready
    lib = go.golib()
    if not _tree_ready:
        for name, (res, args) in TREE_SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _tree_ready = True
    return lib


def search_params(**kw):
    p = SearchParams()
    treelib().bk_search_params_default(ctypes.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise TypeError(f"unknown search parameter {k}")
        if k == "request_steps":
            v = (ctypes.c_int32 * 3)(*(list(v) + [0, 0, 0])[:3])
        setattr(p, k, v)
    if p.request_tasks > 0 and p.request_steps[0] == 0:
        p.request_steps[0] = p.request_tasks
    return p


def default_threads(n_games):
    """Host threads of a pool: the team's workers spin between the steps of a generation (bk_tree.cpp, Team), so they must
    fit the CPUs this process really owns -- the cgroup quota, not the affinity mask -- with room left for the Python thread
    and the HIP runtime's: 12 of a 16-CPU share (16 threads: 25.6 k games/min, erratic, against 27.6 k with 12), and no more
    than one per 8 games of the pool: a rank's share of configs[3] at 8 / 4 / 2 / 1 ranks (two pools of 32 / 64 / 128 / 256
    games) is fastest with 4 / 4-8 / 8 / 8-12 threads and flat beyond (64 games: 0.283 s with 1 thread, 0.221 with 2, 0.203
    with 4 to 8; profiles/r03_shard_of_8_sweep.txt)."""
    own = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            own = min(own, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return max(1, min(12, own - 4 if own > 8 else own // 2, max(1, n_games // 8)))


class GamePool:
    """Lock-step pool of independent self-play games on the native tree core."""

    def __init__(self, seeds, params, cap=4096, threads=None):
        self._lib = treelib()
        self.seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        self.n = len(self.seeds)
        self.cap = max(int(cap), 82)
        threads = threads or default_threads(self.n)
        self._h = self._lib.bk_pool_create(self.n, ctypes.byref(params), self.seeds.ctypes.data, int(threads))
        if not self._h:
            raise RuntimeError("bk_pool_create failed")
        self._feats = self._recs = None   # allocated by the collect flavour in use

    def close(self):
        if getattr(self, "_h", None):         # (a constructor that failed before the handle existed has nothing to close)
            self._lib.bk_pool_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def game_stats(self, g):
        """Visit / value statistics of the moves game g has chosen (bk_pool_game_stats): (root_visits[81] uint64 array,
        sum_root_value, sum_abs_root_value, n_root_values)."""
        st = GameStats()
        if self._lib.bk_pool_game_stats(self._h, int(g), ctypes.byref(st)):
            raise IndexError(g)
        return np.frombuffer(st.root_visits, np.uint64).copy(), st.sum_root_value, st.sum_abs_root_value, int(st.n_root_values)

    def snapshot(self, g=0):
        """The search state of game g as bytes (bk_pool_snapshot: tree, statistics, priors, moves, generator; no nets).
        Only between a deliver and the next collect."""
        n = self._lib.bk_pool_snapshot(self._h, int(g), None, 0)
        if n < 0:
            raise RuntimeError("bk_pool_snapshot: the game has a request out" if n == -2 else f"bk_pool_snapshot failed ({n})")
        buf = ctypes.create_string_buffer(n)
        if self._lib.bk_pool_snapshot(self._h, int(g), buf, n) != n:
            raise RuntimeError("bk_pool_snapshot failed")
        return buf.raw

    def restore(self, g, blob):
        """Replace game g by a snapshot (bk_pool_restore); ValueError if the bytes are not one of this build's."""
        rc = self._lib.bk_pool_restore(self._h, int(g), blob, len(blob))
        if rc == -3:
            raise ValueError("not a snapshot taken by this build of libbkgo.so")
        if rc:
            raise RuntimeError("bk_pool_restore: the game has a request out" if rc == -2 else f"bk_pool_restore failed ({rc})")

    def set_task_cap(self, tasks):
        """Soft limit of a batch in network tasks (bk_pool_set_task_cap); 0: none."""
        self._lib.bk_pool_set_task_cap(self._h, int(tasks))

    def set_dedup(self, on=True):
        """In-batch de-duplication of collect_positions() (bk_pool_set_dedup): equal records travel once."""
        self._lib.bk_pool_set_dedup(self._h, int(bool(on)))

    def set_lanes(self, on=True):
        """Games keep to "their" worker thread (bk_pool_set_lanes; on by default).  Speed only: same games either way."""
        self._lib.bk_pool_set_lanes(self._h, int(bool(on)))

    def dedup_rows(self):
        """(rows the games asked for, rows that travelled) since the pool was created, with de-duplication on."""
        a, b = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._lib.bk_pool_dedup_rows(self._h, ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    def collect(self):
        """-> (uint8 feats [B,27,9,9] view, n_policy); B == 0 when every game is finished."""
        if self._feats is None:
            self._feats = np.empty((self.cap, 27, 9, 9), np.uint8)
        npol = ctypes.c_int(0)
        B = self._lib.bk_pool_collect(self._h, self._feats.ctypes.data, self.cap, ctypes.byref(npol))
        return self._feats[:B], npol.value

    def collect_positions(self):
        """-> (uint8 records [B,192] view, n_policy): the same batch as collect(), as position records whose
        planes the consumer computes (LeafEngine.submit_positions encodes them on the GPU)."""
        if self._recs is None:
            self._recs = np.empty((self.cap, 192), np.uint8)
        npol = ctypes.c_int(0)
        B = self._lib.bk_pool_collect_pos(self._h, self._recs.ctypes.data, self.cap, ctypes.byref(npol))
        return self._recs[:B], npol.value

    def deliver(self, probs, values):
        probs = np.ascontiguousarray(probs, dtype=np.float32)
        values = np.ascontiguousarray(values, dtype=np.float32)
        self._lib.bk_pool_deliver(self._h, probs.ctypes.data, values.ctypes.data)

    @property
    def n_done(self):
        return self._lib.bk_pool_n_done(self._h)

    def info(self, g):
        gi = GameInfo()
        self._lib.bk_pool_game_info(self._h, g, ctypes.byref(gi))
        return {f: getattr(gi, f) for f, _ in GameInfo._fields_ if f != "reserved"}

    def moves(self, g):
        buf = np.empty(128, np.int16)
        n = self._lib.bk_pool_game_moves(self._h, g, buf.ctypes.data, 128)
        return buf[:n].tolist()

    def visits(self, g, ply):
        """{move: N} of the root's children when `ply` was chosen (search_params(record_visits=1))."""
        mv, N = np.empty(81, np.int16), np.empty(81, np.int32)
        n = self._lib.bk_pool_game_visits(self._h, g, ply, mv.ctypes.data, N.ctypes.data)
        if n < 0:
            raise IndexError("no visit record for that ply (record_visits off?)")
        return {int(mv[i]): int(N[i]) for i in range(n)}

    def root_children(self, g):
        mv, N, V = np.empty(81, np.int16), np.empty(81, np.int32), np.empty(81, np.float64)
        n = self._lib.bk_pool_root_children(self._h, g, mv.ctypes.data, N.ctypes.data, V.ctypes.data)
        return {int(mv[i]): (int(N[i]), float(V[i])) for i in range(n)}


def normalise_like_categorical(probs):
    """torch's Categorical(probs) divides by the row sum (reference nnet.py:274): the one-tree search (NativeMCTS, whose
    traces are compared with the reference's) normalises its priors with torch itself."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(probs, dtype=np.float32))
    return (t / t.sum(-1, keepdim=True)).numpy()


def normalise_rows(probs):
    """The same division with the row sum taken left to right in fp32 (bk_normalise_rows): what the lock-step pools use --
    in the Python step loop, in the native one (bk_pools_run) and in examples/bk_selfplay.c -- so that the three play the
    same games bit for bit (torch's sum is vectorised, its order depends on the CPU it runs on)."""
    a = np.array(probs, dtype=np.float32, order="C", copy=True)
    if a.size:
        treelib().bk_normalise_rows(a.ctypes.data, int(a.shape[0]))
    return a


class EngineEvaluator:
    """feats (or position records) -> (probs, values) on a LeafEngine, synchronous or split into submit/finish.
    gpu_encode=True (default): drivers hand over 192-byte position records (GamePool.collect_positions) and
    the engine computes the 27 planes on the GPU; False: host-encoded uint8 planes (GamePool.collect)."""

    def __init__(self, engine, gpu_encode=True, value=True):
        """value=False: an engine without a ValueNet (a tree in simulation mode with policy_net only, mcts.py:68-69): only the
        policy is computed; the values handed back are zeros nobody reads."""
        self.engine = engine
        self.wants_positions = bool(gpu_encode)
        self.value = bool(value)
        self.positions = self.batches = 0

    def submit(self, feats, n_policy):
        self.positions += len(feats)
        self.batches += 1
        if not self.value and n_policy != len(feats):
            raise ValueError("value rows in a request for an engine without a ValueNet")
        if feats.ndim == 2:   # [B,192] position records
            return self.engine.submit_positions(feats, logits=False, probs=n_policy > 0, value=self.value, n_policy=n_policy), (n_policy, len(feats))
        return self.engine.submit(feats, logits=False, probs=n_policy > 0, value=self.value, n_policy=n_policy), (n_policy, len(feats))

    def finish(self, handle, normalise=normalise_like_categorical):
        t, (npol, rows) = handle
        out = self.engine.wait(t)
        probs = normalise(out["probs"]) if npol else np.zeros((0, 81), np.float32)
        return probs, out["value"] if self.value else np.zeros(rows, np.float32)

    def __call__(self, feats, n_policy):
        return self.finish(self.submit(feats, n_policy))


class CallableEvaluator:
    """Any pair of callables policy(x)->logits, value(x)->[B] (CPU tests use the oracle nets)."""

    def __init__(self, policy_fn, value_fn):
        self.policy_fn, self.value_fn = policy_fn, value_fn
        self.positions = self.batches = 0

    def submit(self, feats, n_policy):
        return feats.copy(), n_policy

    def finish(self, handle, normalise=normalise_like_categorical):
        import torch
        feats, npol = handle
        self.positions += len(feats)
        self.batches += 1
        x = feats.astype(np.float32)
        if npol:
            lg = torch.from_numpy(np.asarray(self.policy_fn(x[:npol]), dtype=np.float32))
            probs = normalise(torch.softmax(lg, dim=1).numpy())
        else:
            probs = np.zeros((0, 81), np.float32)
        if self.value_fn is None:          # no value net (simulation mode, mcts.py:68-69): zeros nobody reads
            return probs, np.zeros(len(x), np.float32)
        return probs, np.asarray(self.value_fn(x), dtype=np.float32).reshape(-1)

    def __call__(self, feats, n_policy):
        return self.finish(self.submit(feats, n_policy))


class RecordEvaluator(CallableEvaluator):
    """CallableEvaluator fed with position records (as the engine is: collect_positions): the planes are encoded on the host."""
    wants_positions = True

    def submit(self, recs, n_policy):
        lib = go.golib()
        r = np.ascontiguousarray(recs, np.uint8)
        feats = np.empty((len(r), 27, 9, 9), np.uint8)
        for i in range(len(r)):
            lib.bk_pos_features_u8(ctypes.cast(r[i].ctypes.data, ctypes.POINTER(go.Pos)), feats[i].ctypes.data, 0)
        return feats, n_policy


def callback_evaluator(evaluator):
    """Any Python evaluator (submit / finish) as a bk_evaluator for the native step loop: the CPU tests run bk_pools_run on
    the oracle nets this way.  The batch is evaluated inside submit(); keep the returned struct alive while it is in use."""
    state = {"ticket": 0, "error": None}

    def submit(_ctx, recs, B, n_policy, probs, values):
        try:
            r = np.ctypeslib.as_array((ctypes.c_uint8 * (B * 192)).from_address(recs)).reshape(B, 192)
            if not getattr(evaluator, "wants_positions", False):
                raise TypeError("the native step loop hands over position records: the evaluator must take them (RecordEvaluator)")
            p, v = evaluator.finish(evaluator.submit(r, n_policy), normalise=lambda x: x)
            if n_policy:
                np.ctypeslib.as_array((ctypes.c_float * (n_policy * 81)).from_address(probs))[:] = np.asarray(p, np.float32).reshape(-1)
            np.ctypeslib.as_array((ctypes.c_float * B).from_address(values))[:] = np.asarray(v, np.float32).reshape(-1)
            state["ticket"] += 1
            return state["ticket"]
        except BaseException as exc:          # an exception must not unwind through the C loop
            state["error"] = exc
            return -2

    ev = EvaluatorStruct(None, _SUBMIT_FN(submit), _WAIT_FN(lambda _ctx, _t: 0))
    ev._state = state
    return ev


def run_pools_native(pools, evaluator, cap=None):
    """run_pools with the step loop in C (bk_pools_run): no interpreter between two steps.  evaluator: an EngineEvaluator on
    position records (the engine's own bk_evaluator is used), or any evaluator of records (through callbacks).  Returns the
    number of steps; evaluator.positions / .batches are kept up to date."""
    lib = treelib()
    eng = getattr(evaluator, "engine", None)
    if eng is not None and getattr(evaluator, "wants_positions", False) and getattr(evaluator, "value", True) == eng.has_value:
        ev = eng.evaluator()
    else:
        ev = callback_evaluator(evaluator)
    handles = (ctypes.c_void_p * len(pools))(*[p._h for p in pools])
    info = RunInfo()
    cap = int(cap or min(p.cap for p in pools))
    rc = lib.bk_pools_run(handles, len(pools), ctypes.byref(ev), cap, ctypes.byref(info))
    err = getattr(ev, "_state", {}).get("error")
    if err is not None:
        raise err
    if rc:
        msg = eng._lib.bk_last_error(eng._h).decode() if eng is not None else ""
        raise RuntimeError(f"bk_pools_run failed ({rc}) {msg}")
    if eng is not None and hasattr(evaluator, "positions"):
        evaluator.positions += int(info.rows)
        evaluator.batches += int(info.steps)
    run_pools_native.last_info = {f: getattr(info, f) for f, _ in RunInfo._fields_}
    return int(info.steps)


def run_pools(pools, evaluator, progress=None):
    """Drive one or two pools to completion.  With two pools the host advances one while the
    evaluator (GPU) works on the other's batch.  (The step loop in Python; run_pools_native is the same loop in C.)"""
    inflight = [None] * len(pools)
    live = [True] * len(pools)
    steps = 0
    while any(live) or any(h is not None for h in inflight):
        for i, pool in enumerate(pools):
            if inflight[i] is not None:
                probs, values = evaluator.finish(inflight[i], normalise=normalise_rows)
                pool.deliver(probs, values)
                inflight[i] = None
            if live[i]:
                feats, npol = pool.collect_positions() if getattr(evaluator, "wants_positions", False) else pool.collect()
                if len(feats) == 0:
                    live[i] = False
                else:
                    inflight[i] = evaluator.submit(feats, npol)
                    steps += 1
        if progress is not None:
            progress(steps, pools)
    return steps


# The vector the generation's one all-reduce sums (north_star: "all-reduce visit/value statistics at the end of a generation";
# SURVEY 8e).  Scalars, then two histograms over the 81 points:
#   sum_root_value / sum_abs_root_value / n_root_values   over every ply of every game: V[root] / N[root] when the move was
#       chosen (2 winrate - 1 from the side to move, mcts.py:159-170), its magnitude, and the number of plies
#   first_move_hist[81]      games by their first move
#   root_visit_hist[81]      sum over every ply of every game of the root's children's visit counts, by move (mcts.py:122-128)
# Every entry is an integer or a multiple of 2^-32: a float64 sum of such numbers is exact -- and so independent of the order of
# addition (world size, ring order) -- as long as its magnitude stays below 2^53 * 2^-32 = 2^21.  The counters are far from that;
# the two value sums grow by at most 1 per ply, so the statement holds for generations of fewer than 2^21 = 2,097,152 plies (about
# 26,000 games; configs[3] has 512): named_stats() says so in "value_sums_exact", and EXACT_PLIES is the bound.  173 doubles = 1,384 bytes.
EXACT_PLIES = 1 << 21
STATS_FIELDS = ["games", "black_wins", "white_wins", "plies", "sum_score", "value_evals", "policy_evals", "requests",
                "sum_root_value", "sum_abs_root_value", "n_root_values"]
STATS_HISTS = ["first_move_hist", "root_visit_hist"]
STATS_LEN = len(STATS_FIELDS) + 81 * len(STATS_HISTS)


def pool_stats(pools):
    s = np.zeros(STATS_LEN, np.float64)
    first, visits = len(STATS_FIELDS), len(STATS_FIELDS) + 81
    for p in pools:
        for g in range(p.n):
            gi = p.info(g)
            mv = p.moves(g)
            rv, sv, sa, nv = p.game_stats(g)
            s[0] += 1
            s[1] += gi["score"] > 0
            s[2] += gi["score"] <= 0
            s[3] += gi["n_moves"]
            s[4] += gi["score"]
            s[5] += gi["n_value_evals"]
            s[6] += gi["n_policy_evals"]
            s[7] += gi["n_requests"]
            s[8] += sv
            s[9] += sa
            s[10] += nv
            if mv and mv[0] >= 0:
                s[first + mv[0]] += 1
            s[visits:visits + 81] += rv.astype(np.float64)
    return s


def named_stats(total):
    """The reduced vector as a dict: the scalars by name, the two histograms as integer lists."""
    named = {k: float(total[i]) for i, k in enumerate(STATS_FIELDS)}
    for j, h in enumerate(STATS_HISTS):
        at = len(STATS_FIELDS) + 81 * j
        named[h] = [int(round(x)) for x in total[at:at + 81]]
    named["value_sums_exact"] = bool(named["n_root_values"] < EXACT_PLIES)    # (beyond: still right to ~1e-16 relative, but order-dependent)
    named["mean_root_value"] = named["sum_root_value"] / named["n_root_values"] if named["n_root_values"] else 0.0
    named["mean_abs_root_value"] = named["sum_abs_root_value"] / named["n_root_values"] if named["n_root_values"] else 0.0
    return named


def all_reduce_stats(stats, device=None, native_comm=None, timing=None):
    """The generation's single collective: sum the statistics vector over ranks (no process group: no-op).
    native_comm: a bokego_amd.comm.NativeComm -- the same all-reduce through libbkcomm.so instead of torch.
    Returns (the sums, seconds of the COLLECTIVE).  The ranks finish their games at different times, so a clock started at the call
    would mostly measure the wait for the slowest rank (6 ranks on one card: 0.75 ms on the last rank to arrive, 485 ms on the
    first): every rank first waits at a barrier -- timing["wait_s"], the skew -- and the all-reduce is timed alone behind it."""
    if native_comm is not None:
        t0 = time.perf_counter()
        native_comm.barrier()
        t1 = time.perf_counter()
        out = native_comm.allreduce_sum(stats)
        t2 = time.perf_counter()
        if timing is not None:
            timing["wait_s"] = t1 - t0
        return out, t2 - t1
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        if timing is not None:
            timing["wait_s"] = 0.0
        return stats, 0.0
    t = torch.from_numpy(stats.copy())
    if device is not None:
        t = t.to(device)
    t0 = time.perf_counter()
    dist.barrier()
    if t.is_cuda:
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if t.is_cuda:
        torch.cuda.synchronize()
    t2 = time.perf_counter()
    if timing is not None:
        timing["wait_s"] = t1 - t0
    return t.cpu().numpy(), t2 - t1


def broadcast_weights(policy_sd, value_sd, src=0, device=None, native_comm=None):
    """Generation start (optional): every rank gets rank `src`'s PolicyNet / ValueNet tensors -- one
    broadcast of the two nets' tensors packed back to back (2 x 3.9 MB), through torch.distributed (RCCL on GPUs, gloo
    in the CPU tests) or libbkcomm.so.  Returns (policy_sd, value_sd) as ordered dicts of float32 arrays with the
    caller's names and shapes; without a process group / communicator the inputs come back unchanged.  The reference
    shares its weights between worker processes with share_memory() instead (bin/selfplay.py:171-175)."""
    from collections import OrderedDict
    sds = [OrderedDict((k, np.ascontiguousarray(np.asarray(v.detach().cpu() if hasattr(v, "detach") else v), np.float32))
                       for k, v in sd.items() if not k.endswith("num_batches_tracked")) for sd in (policy_sd, value_sd)]
    flat = np.concatenate([a.reshape(-1) for sd in sds for a in sd.values()])
    if native_comm is not None:
        flat = native_comm.broadcast_f32(flat, src)
    else:
        import torch
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return sds[0], sds[1]
        t = torch.from_numpy(flat)
        if device is not None:
            t = t.to(device)
        dist.broadcast(t, src=src)
        flat = t.cpu().numpy()
    out, at = [], 0
    for sd in sds:
        o = OrderedDict()
        for k, a in sd.items():
            o[k] = flat[at:at + a.size].reshape(a.shape).copy()
            at += a.size
        out.append(o)
    return out[0], out[1]


# self-play: children evaluated at an expansion = the best by prior (bk_search_params.eager_top); in lock-step pools the
# values a game needs later cost no latency of their own -- they ride in the next batch.  The search visits a node's children
# in prior order and ever visits 4 of ~75 in the median (10 at the 90th percentile), so evaluating all of them at the expansion
# (rounds 1-2) was 91 % waste: 3.43 M evaluations per 512-game generation against 0.70 M (4 children) / 0.99 M (8).  Measured on
# one MI355X, 512 games x 400 rollouts/move, the same 512 games move for move (profiles/r03_eager_top.txt):
#   fp32   all 8.4 k games/min | 8: 23.2 k | 6: 24.2 k | 4: 25.3 k | 3: 25.9 k | 2: 24.4 k | 1: 22.6 k    (GPU-bound -> host-bound)
#   f16x2  all 28.1 k          | 4: 35.2 k | 8: 38.4 k | 16: 38.1 k | 24: 35.6 k                          (host-bound throughout)
# and with the host side of the pools reworked (worker team, flat position table, 12 threads): fp32 27.6 k (GPU-bound again: one
# round of workgroups per step), f16x2 59.3 k at 4 children and two pools (8 children / three pools: 53.0 k); and with fp32
# batches held to whole rounds of workgroups (task_cap below): fp32 3: 32.3 k | 4: 32.9 k | 5: 30.8 k | 6: 28.8 k | 8: 25.2 k;
# f16x2 3: 55.4 k | 4: 58.9 k | 6: 62.6 k.
# Round 4 (the one-board workgroups got faster, 334 -> 298 us, and equal rows of a batch travel once): with TWO children per
# expansion a 256-game pool asks for ~480 tasks per step -- one round of 2-board workgroups (534 us) instead of one of 3-board
# ones four fifths full (750 us) -- and the generation needs 17 % fewer evaluations (0.58 M): fp32 512 games 34.6 k -> 36.6 k
# games/min, 256 games (a rank's share at 2 ranks) 27.6 k -> 31.5 k; 128 games the same (23.3 k / 23.5 k), 64 games slower
# (20.9 k -> 19.8 k: there a step is a cooperative launch whose time hardly depends on its size, and two children mean more
# steps).  So fp32 evaluates 2 children from 192 games per rank, else 4 (profiles/r04_eager_probe.txt); 3 is worse than both
# everywhere (its batches straddle the rounds).
EAGER_TOP = {"f32": 4, "f16x2": 6}
EAGER_TOP_F32_MANY_GAMES = (192, 2)       # from this many games per rank: this many children


def small_shard_defaults(precision, eager_top, pool_games):
    """(speculate, task_cap) for a pool of `pool_games` games on an fp32 engine: (0, 0) except where a pool's request (~3 tasks
    per game and step with 4 children per expansion) falls into the 2-CUs-per-board launch's range, 81..128 tasks, whose time
    (174 us) does not depend on the size: there a leaf that reaches 70 visits sends its policy row -- and then its best-prior
    children -- along with requests that go out anyway (bk_search_params.speculate: its expansion at 100 visits needs no round
    trip of its own; the same search) and the batch is held to 128 tasks.  Measured on one MI355X, a rank's 64 games of the
    512-game job (two pools of 32): 1,085 steps of ~85 rows -> 1,009 of ~98, 0.181 -> 0.171 s (profiles/r05_spec_probe.txt; without
    the cap the batches cross 128 tasks: 0.190 s; 128 games per rank, two pools of 64: no gain, left off)."""
    if precision == "f32" and eager_top and eager_top > 2 and 64 < 3 * pool_games <= 128:
        return 70, 128
    return 0, 0


def default_eager_top(precision, n_games_here):
    if precision == "f32" and n_games_here >= EAGER_TOP_F32_MANY_GAMES[0]:
        return EAGER_TOP_F32_MANY_GAMES[1]
    return EAGER_TOP.get(precision, 8)
# In-batch de-duplication of the pools' requests (bk_pool_set_dedup; the in-batch part of the reference's class-level memo,
# mcts.py:41-44).  Measured on one MI355X, 512 games x 400 rollouts/move, the same games move for move (tools/dedup_ab.py,
# profiles/r04_dedup.txt): 7.1 % of the rows are byte-for-byte repeats of another game's row in the same batch (all of them in
# the first plies) -- fp32, which is GPU-bound, 33.2 k -> 34.6 k games/min (+4.2 %; 64 games: +2.0 %); f16x2, which is
# host-bound, pays for the serial hashing with 81 k -> 55 k.  So: on for fp32, off for f16x2.  BK_DEDUP=0/1 overrides.
DEDUP = {"f32": True, "f16x2": False}


NATIVE_LOOP = os.environ.get("BK_NATIVE_LOOP", "1") != "0"     # the pools' step loop in C (bk_pools_run); 0: in Python


def shard_game_ids(n_games, rank, world):
    return [g for g in range(n_games) if g % world == rank]


def self_play(evaluator, n_games=512, rollouts=400, rank=0, world=1, seed_base=20260, noise_weight=0.25,
              sample_plies=8, expand_thresh=100, max_turns=80, cap=4096, threads=None, n_pools=None,
              reduce_device=None, progress=None, prune=1, record_visits=0, native_comm=None, gids=None, eager_top=None, task_cap=None,
              dedup=None, native_loop=None, pool_sizes=None, speculate=None, speculate_rows=8, leaves=1, leaves_visit_only=0):
    """Play this rank's share of a generation; returns (local result dict, reduced stats dict).
    gids: play exactly these game ids instead of the shard `gid % world == rank` -- a game is a pure function of
    `seed_base + gid` and the networks, so the shard of a rank that died can be re-played anywhere (by a survivor, or by
    a later job) with the same games coming out; the reference's launcher can only raise when a worker fails
    (bin/selfplay.py:196-199).  The statistics returned are those of the games played here.
    native_loop: the step loop in C (run_pools_native) instead of in Python (run_pools); the same games either way.
    Default: C whenever the evaluator takes position records and nobody asked for per-step progress calls.
    pool_sizes: how many of this rank's games each pool gets (a list summing to the rank's games; task_cap may then be a list
    too, one per pool).  Default: n_pools equal pools (unequal ones, sized so that each pool's requests land on the cheap side
    of the launch forms' size steps, were measured and gain nothing: profiles/r05_pool_split.txt).
    speculate: evaluation ahead of expansion in the pools (bk_search_params.speculate; None: small_shard_defaults()).
    leaves: > 1 = the OPT-IN multi-leaf throughput mode (bk_search_params.leaves: up to that many rollouts of a step wait for a
    value together, under virtual loss).  Not the reference's search -- other trees, other games, no parity claim --
    but still a pure function of the seeds: the games do not depend on sharding, pools, threads or batch grouping.  Default 1: off."""
    gids = shard_game_ids(n_games, rank, world) if gids is None else [int(g) for g in gids]
    precision = getattr(getattr(evaluator, "engine", None), "precision", "f16x2")
    leaves = max(1, int(leaves))
    if eager_top is None:
        # (multi-leaf mode: the waiting rollouts bring the rows; two children per expansion -- what 192+ games per rank run with --
        # from any share: tools/leaves_cap_sweep.py, 64 games 0.137 -> 0.129 s, 128 games 0.244 -> 0.235)
        eager_top = 2 if (leaves > 1 and precision == "f32") else default_eager_top(precision, len(gids))
    if n_pools is None:
        # two pools: the host advances one while the GPU evaluates the other's batch.  (Rounds 1-2, every child evaluated: three
        # pools paid off for f16x2 from ~200 games per rank; with the best-prior children only, batches are 5x smaller and
        # two fuller ones win in both precisions: f16x2 59.3 k against 51.5 k games/min, fp32 25.9 k against 19.7 k)
        n_pools = 2
    n_pools = max(1, min(n_pools, len(gids))) if gids else 0
    biggest_pool = max(pool_sizes) if pool_sizes else -(-len(gids) // max(1, n_pools))
    small = small_shard_defaults(precision, eager_top, biggest_pool)
    if leaves > 1:
        small, speculate = (0, 0), 0              # (the mode brings its own rows: no evaluation ahead, batches held to whole rounds)
    if speculate is None:
        speculate = small[0]
    prm = search_params(rollouts=rollouts, expand_thresh=expand_thresh, noise_weight=noise_weight,
                        sample_plies=sample_plies, max_turns=max_turns, prune=prune, record_visits=record_visits,
                        eager_top=eager_top, speculate=speculate, speculate_rows=speculate_rows, leaves=leaves,
                        leaves_visit_only=int(bool(leaves_visit_only)))
    if pool_sizes is not None:
        pool_sizes = [int(k) for k in pool_sizes if int(k) > 0]
        if sum(pool_sizes) != len(gids):
            raise ValueError(f"pool_sizes {pool_sizes} do not add up to this rank's {len(gids)} games")
        n_pools, parts, at = len(pool_sizes), [], 0
        for k in pool_sizes:
            parts.append(gids[at:at + k])
            at += k
    else:
        parts = [gids[i::n_pools] for i in range(n_pools)]
    pools = [GamePool([seed_base + g for g in part], prm, cap=cap, threads=threads) for part in parts]
    if task_cap is None:
        # fp32 engine: a step's launch costs whole rounds of 3-board workgroups (0.75 ms per 768 tasks on 256 CUs), and a batch
        # just over a round pays for two: hold batches to the whole number of rounds nearest to what the pool's games ask for
        # (~3 tasks per game and step with 4 children per expansion).  f16x2 steps are host-bound: no limit.
        # (minus 4: policy rows and value rows are rounded up to whole 3-board workgroups separately, and 257 workgroups are
        # two rounds -- rocprofv3 showed 605 of 1,160 launches at exactly 768 tasks taking the 1.0 ms three-round one-board form)
        n_cu = getattr(getattr(evaluator, "engine", None), "n_cu", 256)
        per_round, biggest = 3 * n_cu, max((len(part) for part in parts), default=0)     # (a rank may have no game at all)
        if leaves > 1 and precision == "f32" and eager_top:
            # measured (tools/leaves_cap_sweep.py): a pool's multi-leaf requests come to ~4-6 tasks per game and step; held to whole
            # rounds of one-board workgroups (252 tasks: 296 us) a 32-game pool's steps cost 0.123 s per 64-game share against 0.128
            # uncapped, a 64-game pool's 0.216-0.228 per 128 games against 0.235
            task_cap = n_cu * max(1, round(4.0 * biggest / n_cu)) - 4
        elif precision == "f32" and 0 < eager_top <= 2:
            # ~2 tasks per game and step: whole rounds of 1-, 2- or 3-board workgroups (256 / 512 / 768 tasks on 256 CUs)
            task_cap = n_cu * max(1, round(2.0 * biggest / n_cu)) - 4
        elif small[1] and speculate:
            task_cap = small[1]                   # a 22..42-game pool, evaluation ahead on: requests stay within the 2-CUs-per-board launch
        elif precision == "f32" and eager_top and n_cu == 256 and 128 < 3 * biggest <= 192:
            # a 64-game pool (a rank's share at 4 ranks) asks for ~190 tasks per step: within the range where groups of three
            # boards share 4 CUs (249 us) instead of just over it (a round of one-board workgroups, 298 us): 0.315 -> 0.304 s
            task_cap = 188
        else:
            task_cap = per_round * max(1, round(3.0 * biggest / per_round)) - 4 if (precision == "f32" and eager_top) else 0
    if dedup is None:
        dedup = os.environ["BK_DEDUP"] == "1" if "BK_DEDUP" in os.environ else DEDUP.get(precision, False)
    caps = list(task_cap) if isinstance(task_cap, (list, tuple)) else [task_cap] * len(pools)
    for pool, tc in zip(pools, caps):
        pool.set_task_cap(tc)
        pool.set_dedup(dedup and getattr(evaluator, "wants_positions", False))
    if native_loop is None:
        native_loop = NATIVE_LOOP and progress is None and getattr(evaluator, "wants_positions", False)
    t0 = time.perf_counter()
    if native_loop and pools:
        steps = run_pools_native(pools, evaluator, cap=cap)
    else:
        steps = run_pools(pools, evaluator, progress)
    dt = time.perf_counter() - t0
    games, visits = {}, {}
    for part, pool in zip(parts, pools):
        for i, g in enumerate(part):
            games[g] = {"moves": pool.moves(i), "score": pool.info(i)["score"]}
            if record_visits:
                visits[g] = [pool.visits(i, ply) for ply in range(len(games[g]["moves"]))]
    local = pool_stats(pools) if pools else np.zeros(STATS_LEN)
    rows_req, rows_sent = (sum(x) for x in zip(*[p.dedup_rows() for p in pools])) if pools else (0, 0)
    timing = {}
    total, t_reduce = all_reduce_stats(local, reduce_device, native_comm, timing)
    for p in pools:
        p.close()
    named = named_stats(total)
    return ({"games": games, "visits": visits, "seconds": dt, "steps": steps, "local_stats": local, "native_loop": bool(native_loop),
             "n_pools": n_pools, "speculate": int(speculate), "leaves": leaves, "task_caps": [int(c or 0) for c in caps], "allreduce_s": t_reduce, "allreduce_wait_s": timing.get("wait_s", 0.0), "dedup": bool(dedup), "rows_requested": rows_req, "rows_sent": rows_sent}, named)


# ---- the reference's policy-vs-policy playouts (bin/selfplay.py:18-57) --------------------------------
POLICY_MAX_TURNS = 70  # selfplay.py:16


def legal_sample(pi, game, device=None):
    """Sample a move from the policy; if it is illegal walk the policy's moves in descending
    probability and take the first legal one; None if there is none (selfplay.py:35-47)."""
    import torch
    from . import nnet
    device = device or torch.device("cpu")
    move = nnet.policy_sample(pi, game, device)
    k, moves = 0, None
    while not game.is_legal(move.item()):
        if k == 0:
            moves = torch.topk(nnet.policy_dist(pi, game, device).probs, k=81).indices
        elif k > 80:
            return None
        move = moves[k]
        k += 1
    return move


def playout(game, pi_1, pi_2, device=None):
    """pi_1 (to move first) against pi_2 until turn > 70 or a side has no legal sample (selfplay.py:18-33)."""
    while True:
        for pi in (pi_1, pi_2):
            if game.turn > POLICY_MAX_TURNS:
                return
            mv = legal_sample(pi, game, device)
            if mv is None:
                return
            game.play_move(mv.item())


def policy_self_play(pi1, pi2, num_games, device=None):
    """The reference's self_play(pi1, pi2, num_games) (selfplay.py:49-57); games are scored by area
    (+1 black / -1 white) instead of by a gnugo subprocess."""
    games, results = [], []
    for _ in range(num_games):
        g = go.Game(moves=[])
        playout(g, pi1, pi2, device)
        games.append(g.moves)
        results.append(1 if g.area_score() > 0 else -1)
    return games, results


def batched_policy_playouts(probs_fn, n_games, seed_base=0, max_turns=POLICY_MAX_TURNS):
    """The same playouts for many games in lock-step: one policy batch per ply instead of one
    forward per move per game.  probs_fn(uint8 feats [B,27,9,9]) -> probs [B,81].  Each game samples
    with its own numpy Generator (seed_base + gid); an illegal sample falls back to the policy's
    descending order, as legal_sample does."""
    games = [go.Game(moves=[]) for _ in range(n_games)]
    rngs = [np.random.default_rng(seed_base + i) for i in range(n_games)]
    live = list(range(n_games))
    while live:
        feats = np.stack([games[i].features_u8() for i in live])
        probs = np.asarray(probs_fn(feats), dtype=np.float64)
        nxt = []
        for row, i in enumerate(live):
            g, p = games[i], probs[row] / probs[row].sum()
            mv = int(rngs[i].choice(81, p=p))
            if not g.is_legal(mv):
                mv = next((int(m) for m in np.argsort(-p, kind="stable") if g.is_legal(int(m))), None)
            if mv is None:
                continue
            g.play_move(mv)
            if g.turn <= max_turns:
                nxt.append(i)
        live = nxt
    return [g.moves for g in games], [1 if g.area_score() > 0 else -1 for g in games]


def write_records(out_dir, games, visits=None, komi=5.5):
    """Self-play records: one SGF per game (reference dialect, go.py:528-564) + games.json with the
    move lists, scores and (when recorded) the per-ply root visit counts."""
    os.makedirs(out_dir, exist_ok=True)
    for gid, g in sorted(games.items()):
        s = g["score"]
        go.write_sgf(g["moves"], os.path.join(out_dir, f"game_{gid:05d}.sgf"), komi=komi, B="boke-amd", W="boke-amd",
                     result=("B+" if s > 0 else "W+") + f"{abs(s)}")
    rec = {str(k): dict(v, visits=(visits or {}).get(k)) for k, v in games.items()}
    with open(os.path.join(out_dir, "games.json"), "w") as f:
        json.dump(rec, f)


def main():
    ap = argparse.ArgumentParser(description="MCTS self-play generation on the HIP engine")
    ap.add_argument("--games", type=int, default=512)
    ap.add_argument("--rollouts", type=int, default=400)
    ap.add_argument("--max-turns", type=int, default=80)
    ap.add_argument("--policy", default=None, help=".pt or .bkw policy weights (default: tests/golden/policy_19.bkw)")
    ap.add_argument("--value", default=None)
    ap.add_argument("--max-batch", type=int, default=8192)
    ap.add_argument("--threads", type=int, default=None)
    ap.add_argument("--pools", type=int, default=None, help="lock-step pools per rank (host/GPU overlap; default 3 from 192 games per rank, else 2)")
    ap.add_argument("--host-encode", action="store_true", help="encode the 27 planes on the host instead of on the GPU")
    ap.add_argument("--out", default=None, help="directory for this rank's records (SGF per game + games.json with visit counts)")
    ap.add_argument("--precision", choices=["f32", "f16x2"], default=None,
                    help="conv arithmetic: f32 (default, the reference's width) or the opt-in split-fp16 fast path")
    ap.add_argument("--eager-top", type=int, default=None, help="children evaluated at an expansion, best priors first (0: all; default 4)")
    ap.add_argument("--task-cap", type=int, default=None, help="soft limit of a batch in network tasks (default: whole rounds of workgroups for fp32, none for f16x2; 0: none)")
    ap.add_argument("--replay-shard", metavar="RANK/WORLD", default=None,
                    help="re-play the games a failed rank owned (e.g. 3/8: the gids with gid %% 8 == 3) in this process")
    args = ap.parse_args()

    import torch
    from .bkw import load_bkw
    from .engine import LeafEngine

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # BK_SELFPLAY_BACKEND=gloo BK_SELFPLAY_DEVICE=0: a rehearsal of the N-rank run on ONE card (RCCL refuses two ranks on one
    # GPU): the same sharding, the same all-reduce of the statistics, over gloo on host tensors
    backend = os.environ.get("BK_SELFPLAY_BACKEND", "nccl")
    if "BK_SELFPLAY_DEVICE" in os.environ:
        local_rank = int(os.environ["BK_SELFPLAY_DEVICE"])
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    on_gpu = torch.device("cuda", local_rank) if backend == "nccl" else None

    def load(path, default):
        path = path or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", default)
        if path.endswith(".bkw"):
            return load_bkw(path)
        ck = torch.load(path, map_location="cpu")
        return ck.get("model_state_dict", ck)

    eng = LeafEngine(load(args.policy, "policy_19.bkw"), load(args.value, "value_synth.bkw"), device_id=local_rank,
                     max_batch=args.max_batch, precision=args.precision)
    ev = EngineEvaluator(eng, gpu_encode=not args.host_encode)
    gids = None
    if args.replay_shard:
        r, w = (int(v) for v in args.replay_shard.split("/"))
        gids = shard_game_ids(args.games, r, w)
    local, total = self_play(ev, n_games=args.games, rollouts=args.rollouts, rank=rank, world=world,
                             max_turns=args.max_turns, cap=args.max_batch, threads=args.threads, n_pools=args.pools,
                             reduce_device=on_gpu, record_visits=int(bool(args.out)), gids=gids,
                             eager_top=args.eager_top, task_cap=args.task_cap)
    secs = local["seconds"]
    if world > 1:
        t = torch.tensor([secs], dtype=torch.float64, device=on_gpu or "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        secs = float(t.item())
    if rank == 0:
        print(json.dumps({"games": total["games"], "games_per_min": total["games"] / secs * 60, "seconds": secs,
                          "n_gpus": world, "rollouts_per_move": args.rollouts, "plies": total["plies"],
                          "leaf_evals": total["value_evals"], "leaf_evals_per_s_per_gpu": total["value_evals"] / secs / world,
                          "black_wins": total["black_wins"], "white_wins": total["white_wins"],
                          "allreduce_ms": local["allreduce_s"] * 1e3, "mean_batch": ev.positions / max(1, ev.batches),
                          # the visit / value statistics the all-reduce carries (STATS_FIELDS / STATS_HISTS)
                          "n_root_values": total["n_root_values"], "mean_root_value": total["mean_root_value"],
                          "mean_abs_root_value": total["mean_abs_root_value"], "root_visit_hist": total["root_visit_hist"],
                          "first_move_hist": total["first_move_hist"]}))
    if args.out:
        write_records(os.path.join(args.out, f"rank{rank}"), local["games"], local["visits"])
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
