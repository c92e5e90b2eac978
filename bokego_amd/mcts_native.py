"""NativeMCTS: the reference's single-tree MCTS surface (bokego/mcts.py:15-255) on the native tree
core (csrc/bk_tree.cpp), for fast `genmove`.

    tree = NativeMCTS(Go_MCTS(), policy_net, value_net, expand_thresh=100)
    tree.rollout(1600); best = tree.choose()          # best.last_move, tree.root, tree.winrate()

The search is the same algorithm as bokego_amd.mcts.MCTS (tests compare them visit for visit); only
the tree lives in C++ and Python sees snapshots: `root` and the nodes returned by `choose()` are
position objects (go.Game subclasses with `make_move`, `_terminal`, `dist`, `winrate`, ...).  The
reference's observable dicts -- `tree.N[node]`, `tree.V[node]`, `tree.Q[node]`, `tree.children[node]`
(mcts.py:46-52; GTP.analyze reads `self.N[n]`, gtp.py:386,395) -- are read-only mapping views that
look a node up by its position in the native tree; `rollout(n, analyze_dict)` fills the caller's dict
with the variations exactly as mcts.py:143-147 does, so `analyze` runs on the native search.
Every keyword of the reference's MCTS runs here: expand_thresh, branch_num, exploration_weight, noise_weight, and no_sim=False
with value_net_weight (simulation mode: policy playouts scored into Q; value_net may then be None) -- see bokego_tree.h.
"""
import ctypes
from collections.abc import Mapping

import numpy as np

from . import go, nnet, selfplay

MAX_TURNS = 80
# children evaluated at an expansion: 0 = every child.  One tree is latency-bound -- a request of 2 tasks and one of 64 cost the
# same ~100 us round trip -- so what counts is the NUMBER of requests, and evaluating only the best-prior children
# (bk_search_params.eager_top, what self-play does) doubles it: 1.71 -> 2.45-2.72 ms/move at 4-16 children
# (profiles/r03_eager_top.txt).  Self-play pools, where the values a game needs later ride in the next batch, use it.
EAGER_TOP = 0


class Position(go.Game):
    """A board position as the tree's callers see it (the subset of Go_MCTS used outside the search)."""
    __slots__ = ("_terminal", "tree")

    @classmethod
    def _from_pos(cls, pos, komi=5.5, tree=None):
        n = object.__new__(cls)
        n._pos, n.moves, n.komi, n.sgf = pos, None, komi, None
        n.tree = tree
        n._terminal = n.is_game_over()
        return n

    def __init__(self, board=go.EMPTY_BOARD, ko=None, turn=0, last_move=None):
        super().__init__(board, ko, last_move, turn)
        self.tree = None
        self._terminal = self.is_game_over()

    def is_game_over(self):
        return self._pos.turn > MAX_TURNS or self._pos.last_move == go.PASS

    def copy(self):
        return self._from_pos(go.Pos.from_buffer_copy(self._pos), self.komi, self.tree)

    def make_move(self, index):
        c = self.copy()
        c.play_move(index)
        c._terminal = c.is_game_over()
        return c

    @property
    def winrate(self):
        return None if self.tree is None else self.tree.winrate(self)

    @property
    def dist(self):
        """The node's move distribution as the search uses it (Go_MCTS.dist, mcts.py:371-383): a Categorical over the 81
        points once the policy has been evaluated for this node, else None."""
        return None if self.tree is None else self.tree._dist(self)

    @property
    def value(self):
        return None if self.tree is None else self.tree._value(self)


class _TreeView(Mapping):
    """Read-only dict-like view of one statistic of the native tree, keyed by position (any go.Game): what the
    reference exposes as the defaultdicts MCTS.N / MCTS.V / MCTS.Q (a position the tree has never seen reads 0)."""

    def __init__(self, tree, field):
        self._tree, self._field = tree, field

    def _read(self, i):
        if self._field == "Q":      # summed playout rewards: 0 everywhere unless the tree simulates (no_sim=False)
            q = ctypes.c_double()
            self._tree._lib.bk_pool_node_q(self._tree._pool._h, 0, int(i), ctypes.byref(q))
            return q.value
        return getattr(self._tree._node_at(i), self._field)

    def __getitem__(self, node):
        i = self._tree._find(node)
        return 0 if i < 0 else self._read(i)

    def get(self, node, default=0):
        i = self._tree._find(node)
        return default if i < 0 else self._read(i)

    def __contains__(self, node):
        return self._tree._find(node) >= 0

    def __len__(self):
        return self._tree._pool.info(0)["n_nodes"]

    def __iter__(self):
        return (self._tree._position(i) for i in range(len(self)))


class _ChildrenView(Mapping):
    """MCTS.children: node -> set of child nodes, for expanded nodes (mcts.py:52,185-192)."""

    def __init__(self, tree):
        self._tree = tree

    def __getitem__(self, node):
        i = self._tree._find(node)
        if i < 0 or not self._tree._node_at(i).flags & 1:
            raise KeyError(node)
        return {self._tree._position(c) for c in self._tree._children_ids(i)}

    def __contains__(self, node):
        i = self._tree._find(node)
        return i >= 0 and bool(self._tree._node_at(i).flags & 1)

    def __len__(self):
        return sum(1 for _ in self)

    def __iter__(self):
        n = self._tree._pool.info(0)["n_nodes"]
        return (self._tree._position(i) for i in range(n) if self._tree._node_at(i).flags & 1)


class NativeMCTS:
    """kwargs as the reference's MCTS: expand_thresh, exploration_weight, noise_weight, device; plus
    `evaluator` (anything with __call__(feats_u8, n_policy) -> (probs, values)) and `max_batch`."""

    def __init__(self, root=None, policy_net=None, value_net=None, **kwargs):
        if policy_net is None and kwargs.get("evaluator") is None:
            raise TypeError("Missing required keywork argument: 'policy_net'")
        self.no_sim = kwargs.get("no_sim", True)
        if value_net is None and self.no_sim and kwargs.get("evaluator") is None:
            raise TypeError("Keyword argument 'value_net' is required for no simulation mode")
        # no_sim=False (boke.py --simulate; mcts.py:58,147-148): rollouts end in a policy playout, bk_search_params.simulate.
        # `has_value`: with an `evaluator` instead of nets, whether it computes values (default True).
        has_value = value_net is not None or (kwargs.get("evaluator") is not None and kwargs.get("has_value", True))
        # mcts.py:62,189-190: children = the legal moves among the policy's top k (bk_search_params.branch_num: an expansion then
        # waits for the node's priors).  None / 0 / >= 81: every legal move.
        self.branch_num = kwargs.get("branch_num")
        self.policy_net, self.value_net = policy_net, value_net
        self.expand_thresh = kwargs.get("expand_thresh", 100)
        self.exploration_weight = kwargs.get("exploration_weight", 4.0)
        self.noise_weight = kwargs.get("noise_weight", 0)
        # mcts.py:65-72
        self.value_net_weight = 1.0 if self.no_sim else (0.0 if not has_value else kwargs.get("value_net_weight", 0.5))
        self._max_batch = kwargs.get("max_batch")
        ev = kwargs.get("evaluator")
        if ev is None:
            ev = self._evaluator_from_nets()
        self.evaluator = ev
        # Evaluation ahead of expansion (bokego_tree.h, `speculate`): same search, fewer round trips, more rows per request.
        # It pays as long as the bigger request costs about the same: the f16x2 kernel runs <= 256 rows as one round of
        # one-CU workgroups (1.85 -> 1.45 ms/move at 1600 rollouts over the first 40 moves); the fp32 kernel's small-batch
        # launch gives a board 4 CUs up to 64 rows, 3 up to 80 but only 2 from 81, so there a request takes speculative rows
        # only while it stays within 80 (1.91 -> 1.79 ms/move over 80-move games; with 128 or 256 rows it gets SLOWER).
        # Both follow the engine's precision when it is switched later (engine.set_precision), unless given explicitly.
        # request_tasks (fp32 engines: 64): requests stay within the range in which the cooperative launch gives a board 4 CUs
        # -- candidates go out in parts, overflowing children of already-evaluated nodes wait for the next request.
        self._spec_kw = (kwargs.get("speculate"), kwargs.get("speculate_rows"), kwargs.get("request_tasks"))
        self._spec_prec = self._engine_precision()
        spec = self._spec_defaults(self._spec_prec)
        prm = selfplay.search_params(rollouts=0, expand_thresh=self.expand_thresh, c_puct=self.exploration_weight,
                                     noise_weight=self.noise_weight, max_turns=MAX_TURNS, prune=kwargs.get("prune", 0),
                                     speculate=spec[0], speculate_rows=spec[1], request_tasks=spec[2],
                                     request_steps=kwargs.get("request_steps", (spec[2], 80, 128) if spec[2] == 64 else (spec[2],)))
        prm.eager_top = kwargs.get("eager_top", EAGER_TOP)
        # OPT-IN, not the reference's search: `leaves` > 1 rollouts of a step wait for their values together under virtual loss
        # (bk_search_params.leaves).  Default 1 = the reference's sequential search, rollout for rollout.
        prm.leaves = max(1, int(kwargs.get("leaves", 1)))
        prm.leaves_visit_only = int(bool(kwargs.get("leaves_visit_only", 0)))
        if prm.leaves > 1:
            prm.speculate, prm.request_tasks = 0, 0
            prm.request_steps[0] = prm.request_steps[1] = prm.request_steps[2] = 0
        prm.simulate, prm.use_value, prm.value_weight = int(not self.no_sim), int(has_value), float(self.value_net_weight)
        if self.branch_num is not None and 0 <= self.branch_num < go.N ** 2:
            if self.branch_num == 0:
                raise NotImplementedError("branch_num = 0 (no children at all) is not a search; use bokego_amd.mcts.MCTS to reproduce it")
            prm.branch_num = int(self.branch_num)
        self._cap = kwargs.get("cap", 1024)
        self._pool = selfplay.GamePool([kwargs.get("seed", 0)], prm, cap=self._cap, threads=1)
        self.N, self.V, self.Q = _TreeView(self, "N"), _TreeView(self, "V"), _TreeView(self, "Q")
        self._has_value = has_value
        self.children = _ChildrenView(self)
        self._lib = self._pool._lib
        self._lib.bk_pool_set_manual(self._pool._h, 1)
        self.komi = getattr(root, "komi", 5.5) if root is not None else 5.5
        if root is not None and root.key() != Position().key():
            self._set_position(root)
        self._pump()

    # ---- plumbing ------------------------------------------------------------------------------------
    def _evaluator_from_nets(self):
        policy_net, value_net = self.policy_net, self.value_net
        if isinstance(policy_net, nnet.HipPolicyNet) and isinstance(value_net, nnet.HipValueNet):
            return selfplay.EngineEvaluator(nnet.fuse(policy_net, value_net, self._max_batch))
        if isinstance(policy_net, nnet.HipPolicyNet) and value_net is None:
            return selfplay.EngineEvaluator(policy_net.engine(), value=False)
        # any callables: policy(x)->logits, value(x)->[B,1]  (CPU tests use the oracle nets)
        import torch
        return selfplay.CallableEvaluator(lambda x: policy_net(torch.from_numpy(x)).numpy(),
                                          None if value_net is None else
                                          (lambda x: value_net(torch.from_numpy(x)).numpy().reshape(-1)))

    # ---- copies and pickles (mcts.py:81-108): the search state travels, the networks do not -----------------
    # what does not travel in a pickle / is not shared by a copy: the nets and the evaluator (mcts.py:93-96), and the handles
    # and views of THIS object's native pool (a snapshot stands in for them)
    _NOT_STATE = ("policy_net", "value_net", "evaluator", "_pool", "_lib", "N", "V", "Q", "children")

    def _adopt(self, blob, cap):
        """a fresh one-game pool holding the snapshot `blob` (its search parameters come with it)"""
        self._pool = selfplay.GamePool([0], selfplay.search_params(rollouts=0), cap=cap, threads=1)
        self._lib = self._pool._lib
        self._pool.restore(0, blob)
        self.N, self.V, self.Q = _TreeView(self, "N"), _TreeView(self, "V"), _TreeView(self, "Q")
        self.children = _ChildrenView(self)

    def __getstate__(self):
        """MCTS.__getstate__ (mcts.py:93-96): everything but the nets -- every attribute of the object (a subclass's too:
        NativeGTP's protocol state) and the native tree as a snapshot (bk_pool_snapshot: nodes, edges, priors, N / V / Q, root,
        generator)."""
        d = {k: v for k, v in self.__dict__.items() if k not in self._NOT_STATE}
        d["snapshot"] = self._pool.snapshot(0)
        return d

    def __setstate__(self, state):
        """MCTS.__setstate__ (mcts.py:97-108): the tree is back; `policy_net` / `value_net` are None and are set by the
        caller (`tree.policy_net = ...; tree.value_net = ...`), the evaluator is rebuilt from them on first use."""
        state = dict(state)
        blob = state.pop("snapshot")
        self.__dict__.update(state)
        self.policy_net = self.value_net = self.evaluator = None
        self._adopt(blob, self._cap)

    def __deepcopy__(self, memo):
        """MCTS.__deepcopy__ (mcts.py:81-90): an independent copy of the search state that shares the networks -- and here
        the evaluator (the engine) -- with the original."""
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k not in self._NOT_STATE:
                setattr(new, k, copy.deepcopy(v, memo))
        new.policy_net, new.value_net, new.evaluator = self.policy_net, self.value_net, self.evaluator
        new._adopt(self._pool.snapshot(0), self._cap)
        return new

    def _ready(self):
        """an unpickled tree gets its evaluator from the nets the caller has put back (mcts.py:106-108); called before anything
        is changed in the native tree, so that a tree without nets refuses the call and stays as it is"""
        if self.evaluator is None:
            if self.policy_net is None:
                raise RuntimeError("this tree was unpickled without its networks: set tree.policy_net / tree.value_net first")
            self.evaluator = self._evaluator_from_nets()

    def _engine_precision(self):
        return getattr(getattr(self.evaluator, "engine", None), "precision", None)

    def _spec_defaults(self, prec):
        d = (50, 256, 0) if prec == "f16x2" else (50, 80, 64) if prec == "f32" else (0, 128, 0)
        return tuple(d[i] if self._spec_kw[i] is None else self._spec_kw[i] for i in (0, 1, 2))

    def _pump(self):
        """Run the native search until it needs nothing more (every outstanding rollout done)."""
        self._ready()
        prec = self._engine_precision()
        if prec != self._spec_prec:           # engine.set_precision() since the last call: the other kernel's defaults
            self._spec_prec = prec
            self._lib.bk_pool_set_speculation(self._pool._h, *self._spec_defaults(prec))
        while True:
            if getattr(self.evaluator, "wants_positions", False):
                feats, npol = self._pool.collect_positions()   # planes are encoded on the GPU
            else:
                feats, npol = self._pool.collect()
            if len(feats) == 0:
                return
            self._pool.deliver(*self.evaluator(feats, npol))

    # node ids <-> positions (ids are valid until the next re-rooting of a pruning tree: never cached here)
    def _find(self, node):
        return self._lib.bk_pool_find(self._pool._h, 0, ctypes.byref(node._pos))

    def _node_at(self, i):
        info = selfplay.NodeInfo()
        if self._lib.bk_pool_node(self._pool._h, 0, int(i), ctypes.byref(info), None):
            raise IndexError(i)
        return info

    def _node_info(self, node):
        i = self._find(node)
        return None if i < 0 else self._node_at(i)

    def _position(self, i):
        pos = go.Pos()
        if self._lib.bk_pool_node(self._pool._h, 0, int(i), None, ctypes.byref(pos)):
            raise IndexError(i)
        return Position._from_pos(pos, self.komi, self)

    def _children_ids(self, i):
        ids = np.empty(81, np.int32)
        n = self._lib.bk_pool_node_children(self._pool._h, 0, int(i), ids.ctypes.data, 81)
        return ids[:max(n, 0)].tolist()

    def _dist(self, node):
        i = self._find(node)
        pr = np.empty(81, np.float64)
        if i < 0 or self._lib.bk_pool_node_prior(self._pool._h, 0, i, pr.ctypes.data):
            return None
        import torch
        t = torch.from_numpy(pr.astype(np.float32))
        d = torch.distributions.Categorical(probs=t, validate_args=False)
        d.probs = t       # exactly what the search uses (the constructor renormalises once more; the reference overwrites
        return d          # dist.probs the same way, mcts.py:357,369,381)

    def _value(self, node):
        if not self._has_value:      # Go_MCTS.value without a value net, mcts.py:395-396
            return None
        info = self._node_info(node)
        return None if info is None or not info.flags & 4 else float(info.value)

    def _set_position(self, node):
        if self._lib.bk_pool_set_position(self._pool._h, 0, ctypes.byref(node._pos)):
            raise RuntimeError("bk_pool_set_position failed")

    # ---- the reference surface ----------------------------------------------------------------------------
    @property
    def root(self):
        pos = go.Pos()
        self._lib.bk_pool_root_pos(self._pool._h, 0, ctypes.byref(pos))
        r = Position._from_pos(pos, self.komi, self)
        return r

    def rollout(self, n=1, analyze_dict=None):
        """n rollouts from the root (mcts.py:133-151).  analyze_dict: as in the reference, every descent longer than two
        nodes is stored under the root child it went through (child -> [child, ..., leaf]); the native search records the
        node ids, the dict receives position objects."""
        self._ready()
        if analyze_dict is not None:
            self._lib.bk_pool_set_analyze(self._pool._h, 1)
        self._lib.bk_pool_add_rollouts(self._pool._h, 0, int(n))
        self._pump()
        if analyze_dict is not None:
            ids = np.empty(128, np.int32)
            for mv in self.child_stats():
                k = self._lib.bk_pool_variation(self._pool._h, 0, int(mv), ids.ctypes.data, 128)
                if k > 0:
                    line = [self._position(i) for i in ids[:min(k, 128)]]
                    analyze_dict[line[0]] = line
            self._lib.bk_pool_set_analyze(self._pool._h, 0)

    def principal_variation(self):
        """Moves of the most visited line from the root."""
        mv = np.empty(128, np.int16)
        n = self._lib.bk_pool_principal_variation(self._pool._h, 0, mv.ctypes.data, 128)
        return mv[:max(n, 0)].tolist()

    def choose(self, node=None):
        """Most visited child of the root becomes the new root; `node`: choose from another node of the tree -- the best
        child is returned and the root stays where it is (mcts.py:110-131)."""
        if node is not None and node.key() != self.root.key():
            if getattr(node, "_terminal", False):
                return node
            i = self._find(node)
            kids = self._children_ids(i) if i >= 0 and self._node_at(i).flags & 1 else []
            if not kids:
                # not expanded (`node not in self.children`): the reference samples a child from the node's policy,
                # mcts.py:119-120 -- Position.find_random_child does that, through this tree's evaluator
                return self._sample_child(self._position(i) if i >= 0 else node)
            best, best_n = None, None
            for c in kids:                     # most visited, unseen children last; first in move order on ties
                cn = self._node_at(c).N
                sc = float("-inf") if cn == 0 else cn
                if best is None or sc > best_n:
                    best, best_n = c, sc
            return self._position(best)
        r = self.root
        if r._terminal:
            return r
        self._ready()
        mv = self._lib.bk_pool_choose(self._pool._h, 0)
        if mv == go._NO_MOVE:
            # the root has no children -- no legal move at all, or (branch_num) none among the policy's top k: the reference's
            # max() over an empty set raises here; like the Python tree, sample a move from the policy (a pass as the last resort)
            child = self._sample_child(r)
            self.play(go.PASS if child.last_move is None else child.last_move)
        self._pump()
        return self.root

    def _sample_child(self, node):
        """Go_MCTS.find_random_child (mcts.py:337-360) for a node of this tree: a legal, non-eye-filling move sampled from the
        node's policy -- the tree's prior where it has one, else one evaluation through the tree's evaluator -- passing as
        the last resort."""
        import torch
        if getattr(node, "_terminal", False):
            return node
        d = self._dist(node)
        if d is None:
            planes = node.features_u8()                  # (refreshes the liberty cache, as the search's requests do)
            if getattr(self.evaluator, "wants_positions", False):
                rows = np.frombuffer(bytes(node._pos), np.uint8).reshape(1, 192).copy()
            else:
                rows = planes[None]
            probs, _ = self.evaluator(rows, 1)
            pr = torch.from_numpy(np.asarray(probs[0], np.float32))
        else:
            pr = d.probs.clone()                         # (the tree's own prior is left alone)
        d = torch.distributions.Categorical(probs=pr, validate_args=False)
        d.probs = pr
        color = 1 if node.turn % 2 == 0 else 2
        move, tries = d.sample().item(), 0
        while not node.is_legal(move) or go.golib().bk_pos_possible_eye(ctypes.byref(node._pos), move) == color:
            d.probs[move] = 0                            # as the reference: a rejected move is not drawn again
            if tries >= go.N ** 2 or not float(d.probs.sum()) > 0:
                move = go.PASS
                break
            move, tries = d.sample().item(), tries + 1
        child = node.copy()
        child.tree = self
        child.play_pass() if move == go.PASS else child.play_move(move)
        child._terminal = child.is_game_over()
        return child

    def play(self, move):
        """An outside move: the root's child for `move` (created if needed) becomes the root."""
        self._ready()
        rc = self._lib.bk_pool_play(self._pool._h, 0, int(move))
        if rc:
            raise go.IllegalMove(self.root, rule_type=go._RULES.get(rc), sq_c=move)
        self._pump()

    def set_root(self, node):
        """Any position; if it is one move away from the current root the subtree is kept."""
        self._ready()
        r = self.root
        self.komi = getattr(node, "komi", self.komi)
        if node.turn == r.turn + 1 and node.last_move is not None:
            probe = go.Pos.from_buffer_copy(r._pos)
            if go.golib().bk_pos_play(ctypes.byref(probe), node.last_move) == 0 and \
                    Position._from_pos(probe).key() == node.key():
                self.play(node.last_move)
                return
        self._set_position(node)
        self._pump()

    def winrate(self, node=None):
        """(((1 - w) Q + w V) / N + 1) / 2 of the root, or of any node of the tree (mcts.py:159-170; w = value_net_weight, 1
        without simulations); 0 for an unvisited one."""
        i = self._lib.bk_pool_root_id(self._pool._h, 0) if node is None else self._find(node)
        if i < 0:
            return 0
        info = self._node_at(i)
        if info.N <= 0:
            return 0
        w = self.value_net_weight
        q = self.Q._read(i) if not self.no_sim else 0.0
        return (((1 - w) * q + w * info.V) / info.N + 1) / 2

    def child_stats(self):
        """{move: (N, V)} of the root's children."""
        return self._pool.root_children(0)

    def clear_cache(self):
        pass  # the native tree is pruned at re-rooting when created with prune=1

    def close(self):
        self._pool.close()
