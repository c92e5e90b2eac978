"""NativeMCTS: the reference's single-tree MCTS surface (bokego/mcts.py:15-255) on the native tree
core (csrc/bk_tree.cpp), for fast `genmove`.

    tree = NativeMCTS(Go_MCTS(), policy_net, value_net, expand_thresh=100)
    tree.rollout(1600); best = tree.choose()          # best.last_move, tree.root, tree.winrate()

The search is the same algorithm as bokego_amd.mcts.MCTS (tests compare them visit for visit); only
the tree lives in C++ and Python sees snapshots: `root` and the nodes returned by `choose()` are
position objects (go.Game subclasses with `make_move`, `_terminal`, ...), and statistics are read
with `child_stats()` / `N` of the root's children rather than through dicts keyed by every node.
"""
import ctypes


from . import go, nnet, selfplay

MAX_TURNS = 80


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


class NativeMCTS:
    """kwargs as the reference's MCTS: expand_thresh, exploration_weight, noise_weight, device; plus
    `evaluator` (anything with __call__(feats_u8, n_policy) -> (probs, values)) and `max_batch`."""

    def __init__(self, root=None, policy_net=None, value_net=None, **kwargs):
        if policy_net is None and kwargs.get("evaluator") is None:
            raise TypeError("Missing required keywork argument: 'policy_net'")
        self.no_sim = kwargs.get("no_sim", True)
        if not self.no_sim:
            raise NotImplementedError("NativeMCTS implements the no-simulation mode only")
        if value_net is None and kwargs.get("evaluator") is None:
            raise TypeError("Keyword argument 'value_net' is required for no simulation mode")
        self.policy_net, self.value_net = policy_net, value_net
        self.expand_thresh = kwargs.get("expand_thresh", 100)
        self.exploration_weight = kwargs.get("exploration_weight", 4.0)
        self.noise_weight = kwargs.get("noise_weight", 0)
        self.value_net_weight = 1.0
        ev = kwargs.get("evaluator")
        if ev is None:
            if isinstance(policy_net, nnet.HipPolicyNet) and isinstance(value_net, nnet.HipValueNet):
                ev = selfplay.EngineEvaluator(nnet.fuse(policy_net, value_net, kwargs.get("max_batch")))
            else:  # any callables: policy(x)->logits, value(x)->[B,1]  (CPU tests use the oracle nets)
                import torch
                ev = selfplay.CallableEvaluator(lambda x: policy_net(torch.from_numpy(x)).numpy(),
                                                lambda x: value_net(torch.from_numpy(x)).numpy().reshape(-1))
        self.evaluator = ev
        # Evaluation ahead of expansion (bokego_tree.h, `speculate`): same search, fewer round trips, more rows per request.
        # It pays as long as the bigger request costs about the same: the f16x2 kernel runs <= 256 rows as one round of
        # one-CU workgroups (1.85 -> 1.45 ms/move at 1600 rollouts over the first 40 moves); the fp32 kernel's small-batch
        # launch gives a board 4 CUs up to 64 rows, 3 up to 80 but only 2 from 81, so there a request takes speculative rows
        # only while it stays within 80 (1.91 -> 1.79 ms/move over 80-move games; with 128 or 256 rows it gets SLOWER).
        prec = getattr(getattr(ev, "engine", None), "precision", None)
        spec_default = (50, 256) if prec == "f16x2" else (50, 80) if prec == "f32" else (0, 128)
        prm = selfplay.search_params(rollouts=0, expand_thresh=self.expand_thresh, c_puct=self.exploration_weight,
                                     noise_weight=self.noise_weight, max_turns=MAX_TURNS, prune=kwargs.get("prune", 0),
                                     speculate=kwargs.get("speculate", spec_default[0]),
                                     speculate_rows=kwargs.get("speculate_rows", spec_default[1]))
        self._pool = selfplay.GamePool([kwargs.get("seed", 0)], prm, cap=kwargs.get("cap", 1024), threads=1)
        self._lib = self._pool._lib
        self._lib.bk_pool_set_manual(self._pool._h, 1)
        self.komi = getattr(root, "komi", 5.5) if root is not None else 5.5
        if root is not None and root.key() != Position().key():
            self._set_position(root)
        self._pump()

    # ---- plumbing ------------------------------------------------------------------------------------
    def _pump(self):
        """Run the native search until it needs nothing more (every outstanding rollout done)."""
        while True:
            if getattr(self.evaluator, "wants_positions", False):
                feats, npol = self._pool.collect_positions()   # planes are encoded on the GPU
            else:
                feats, npol = self._pool.collect()
            if len(feats) == 0:
                return
            self._pool.deliver(*self.evaluator(feats, npol))

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
        self._lib.bk_pool_add_rollouts(self._pool._h, 0, int(n))
        self._pump()

    def choose(self, node=None):
        """Most visited child of the root becomes the new root (mcts.py:110-131)."""
        if node is not None and node.key() != self.root.key():
            raise NotImplementedError("NativeMCTS.choose works at the root")
        r = self.root
        if r._terminal:
            return r
        mv = self._lib.bk_pool_choose(self._pool._h, 0)
        if mv == go._NO_MOVE:      # no legal move at all: pass (the Python tree samples, ending in a pass)
            self.play(go.PASS)
        self._pump()
        return self.root

    def play(self, move):
        """An outside move: the root's child for `move` (created if needed) becomes the root."""
        rc = self._lib.bk_pool_play(self._pool._h, 0, int(move))
        if rc:
            raise go.IllegalMove(self.root, rule_type=go._RULES.get(rc), sq_c=move)
        self._pump()

    def set_root(self, node):
        """Any position; if it is one move away from the current root the subtree is kept."""
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
        """(V/N + 1)/2 of the root (or of one of its children)."""
        if node is None or node.key() == self.root.key():
            gi = self._pool.info(0)
            n, v = gi["root_N"], gi["root_V"]
        else:
            st = self.child_stats().get(node.last_move)
            if st is None or self.root.make_move(node.last_move).key() != node.key():
                return 0
            n, v = st
        return (v / n + 1) / 2 if n > 0 else 0

    def child_stats(self):
        """{move: (N, V)} of the root's children."""
        return self._pool.root_children(0)

    def clear_cache(self):
        pass  # the native tree is pruned at re-rooting when created with prune=1

    def close(self):
        self._pool.close()
