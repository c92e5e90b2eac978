"""PUCT tree search with the call surface of the reference's bokego/mcts.py, refactored to feed
the GPU in batches.

Kept from the reference (mcts.py:15-255, 257-409): `MCTS(root, policy_net, value_net, **kwargs)` with
kwargs no_sim / expand_thresh / branch_num / exploration_weight / noise_weight / value_net_weight /
device; `rollout(n, analyze_dict)`, `choose(node)`, `set_root(node)`, `winrate(node)`; observable
`Q`, `N`, `V`, `children`, `root`; node type `Go_MCTS(go.Game)` with lazy `.features`, `.dist`,
`.value`, `.winrate`, `find_children`, `make_move`, `get_move`, `is_game_over`.  The search itself
is the same sequential PUCT: score(child) = -avg + c * P(move) * sqrt(sum N) / (1 + N), a leaf is
expanded on the visit after N > expand_thresh, the value of the path's last node is backed up with
alternating sign.

What changed is WHEN the networks run.  The reference evaluates one position per call, lazily
(mcts.py:371-403).  Here a node's expansion evaluates the node's policy and the value of every new
child in ONE engine call (pure memoised functions, so the search result is the same),
and the search is written as a generator of evaluation requests so that many trees can be
advanced in lock-step into one queue (bokego_amd/selfplay.py).

Differences a caller can observe: children are kept in ascending move order and ties in
`max()` go to the lowest move index (the reference iterates a Python set whose order depends on
a Zobrist table drawn at import); memo tables are per tree, not class-level.
"""
import ctypes
from math import sqrt

import numpy as np
import torch
from torch.distributions import categorical, dirichlet

from . import go, nnet

MAX_TURNS = 80


class Go_MCTS(go.Game):
    """A search-tree node: go.Game + statistics + memoised network outputs (reference mcts.py:257-409)."""

    __slots__ = ("_terminal", "tree", "N", "V", "Q", "_kids", "_prior", "_value", "_fts", "_dist", "mv")

    def __init__(self, board=go.EMPTY_BOARD, ko=None, turn=0, last_move=None):
        super().__init__(board, ko, last_move, turn)
        self._init_node()

    def _init_node(self):
        self.tree = None
        self.N, self.V, self.Q = 0, 0.0, 0
        self._kids = None      # list of child nodes once expanded
        self._prior = None     # python list of 81 floats (Categorical-normalised policy)
        self._value = None
        self._fts = None       # uint8 (27,9,9)
        self._dist = None
        self.mv = int(self._pos.last_move)
        self._terminal = self.is_game_over()

    @classmethod
    def _from_pos(cls, pos, komi=5.5):
        n = object.__new__(cls)
        n._pos = pos
        n.moves, n.komi, n.sgf = None, komi, None
        n._init_node()
        return n

    def copy(self):
        return self._from_pos(go.Pos.from_buffer_copy(self._pos), self.komi)

    # pickling (reference mcts.py:279-292 ships board/ko/turn/last_move + the liberty cache; the 192-byte record
    # holds exactly that).  Statistics and memoised outputs travel too; the tree link is restored by
    # MCTS.__setstate__, the Categorical and the planes are rebuilt on demand.
    def __getstate__(self):
        return {"pos": bytes(self._pos), "komi": self.komi, "N": self.N, "V": self.V, "Q": self.Q, "kids": self._kids,
                "prior": self._prior, "value": self._value}

    def __setstate__(self, st):
        self._pos = go.Pos.from_buffer_copy(st["pos"])
        self.moves, self.komi, self.sgf = None, st["komi"], None
        self._init_node()
        self.N, self.V, self.Q, self._kids = st["N"], st["V"], st["Q"], st["kids"]
        self._prior, self._value = st["prior"], st["value"]

    def _clone_with_stats(self):
        c = self.copy()
        c.N, c.V, c.Q, c._value, c._fts = self.N, self.V, self.Q, self._value, self._fts
        c._prior = list(self._prior) if self._prior is not None else None
        return c

    def is_game_over(self):
        """Terminal after MAX_TURNS or when the last move was a pass (mcts.py:362-364)."""
        return self._pos.turn > MAX_TURNS or self._pos.last_move == go.PASS

    # ---- children ----------------------------------------------------------------------------------
    def make_move(self, index):
        """Copy of the position after `index` was played (mcts.py:340-346)."""
        c = self.copy()
        c.play_move(index)
        c.mv = int(c._pos.last_move)
        c._terminal = c.is_game_over()
        c.tree = self.tree            # the reference deep-copies the node, tree link included (mcts.py:343)
        return c

    def find_children(self, k=None):
        """All legal successors (or the legal ones among the policy's top k), mcts.py:309-317."""
        if self._terminal:
            return []
        arr = (go.Pos * 81)()
        mvs = (ctypes.c_int16 * 81)()
        n = go.golib().bk_pos_children(ctypes.byref(self._pos), arr, mvs)
        keep = range(n)
        if k is not None and 0 <= k < go.N ** 2:
            top = set(self.topk_moves(k))
            keep = [i for i in range(n) if mvs[i] in top]
        return [self._from_pos(go.Pos.from_buffer_copy(arr[i]), self.komi) for i in keep]

    def find_random_child(self):
        if self._terminal:
            return self
        return self.make_move(self.get_move())

    def topk_moves(self, k):
        return torch.topk(self.dist.probs, k=k).indices.tolist()

    def get_move(self):
        """Sample a legal, non-eye-filling move from the policy; pass as a last resort (mcts.py:348-360)."""
        d = self.dist
        if not float(d.probs.sum()) > 0:           # a node whose moves were all used up by an earlier playout
            return go.PASS
        move = d.sample().item()
        color = 1 if self.turn % 2 == 0 else 2
        tries = 0
        while not self.is_legal(move) or go.golib().bk_pos_possible_eye(ctypes.byref(self._pos), move) == color:
            if tries >= go.N ** 2:
                return go.PASS
            d.probs[move] = 0
            if not float(d.probs.sum()) > 0:       # every move with a non-zero probability is used up: the reference's
                return go.PASS                     # sample() raises here (multinomial of an all-zero row); pass instead
            move = d.sample().item()
            tries += 1
        return move

    def reward(self, gnu=False):
        """1 if Black wins else -1, by area score (gnugo scoring is out of scope)."""
        return 1 if self.score() > 0 else -1

    # ---- memoised network outputs ----------------------------------------------------------------------
    def _add_noise(self, weight):
        """Mix Dirichlet(0.1) noise into the prior (mcts.py:366-369); one RNG draw even for weight 0."""
        noise = MCTS._dirichlet.sample()
        d = self.dist
        d.probs = (1 - weight) * d.probs + weight * noise
        self._prior = d.probs.tolist()

    @property
    def features(self):
        if self._fts is None:
            self._fts = self.features_u8()
        return torch.from_numpy(self._fts.astype(np.float32))

    @property
    def dist(self):
        if self.tree is None:
            return None
        if self._prior is None:
            self.tree._eval_now([self], [])
        if self._dist is None:
            t = torch.tensor(self._prior, dtype=torch.float32)
            self._dist = categorical.Categorical(probs=t, validate_args=False)
            self._dist.probs = t  # already normalised once, like Categorical(SOFT(logits)) in the reference
        return self._dist

    @property
    def value(self):
        if self.tree is None or self.tree.value_net is None:
            return None
        if self._value is None:
            self.tree._eval_now([], [self])
        return self._value

    @property
    def winrate(self):
        if self.tree is None:
            return None
        return self.tree.winrate(self)


class EvalRequest:
    """Positions a tree wants evaluated: policy (+value) for `policy_nodes`, value only for `value_nodes`."""
    __slots__ = ("policy_nodes", "value_nodes")

    def __init__(self, policy_nodes, value_nodes):
        self.policy_nodes, self.value_nodes = policy_nodes, value_nodes

    def __len__(self):
        return len(self.policy_nodes) + len(self.value_nodes)


class Evaluator:
    """Runs EvalRequests of any number of trees as one batch and writes the results into the nodes."""

    def __init__(self, policy_net, value_net, device=torch.device("cpu"), max_batch=None):
        self.policy_net, self.value_net, self.device = policy_net, value_net, device
        self.engine = None
        if isinstance(policy_net, nnet.HipPolicyNet) and isinstance(value_net, nnet.HipValueNet):
            self.engine = nnet.fuse(policy_net, value_net, max_batch)
        self.n_batches = 0
        self.n_positions = 0
        self.n_policy = 0

    @staticmethod
    def _features(nodes):
        f = np.empty((len(nodes), 27, 9, 9), np.uint8)
        for i, n in enumerate(nodes):
            if n._fts is None:
                n._fts = n.features_u8()
            f[i] = n._fts
        return f

    def run(self, requests):
        pol = [n for r in requests for n in r.policy_nodes]
        val = [n for r in requests for n in r.value_nodes]
        nodes = pol + val
        if not nodes:
            return
        f = self._features(nodes)
        want_v = self.value_net is not None
        cap = self.engine.max_batch if self.engine is not None else 1 << 30
        if len(nodes) > cap:  # rare: split, policy nodes stay a prefix of the first chunks
            for i in range(0, len(nodes), cap):
                chunk = nodes[i:i + cap]
                npol = max(0, min(len(pol) - i, len(chunk)))
                self._run_chunk(chunk, f[i:i + cap], npol, want_v)
        else:
            self._run_chunk(nodes, f, len(pol), want_v)

    def _run_chunk(self, nodes, f, npol, want_v):
        self.n_batches += 1
        self.n_positions += len(nodes)
        self.n_policy += npol
        if self.engine is not None:
            out = self.engine.eval(f, logits=False, probs=npol > 0, value=want_v, n_policy=npol)
            probs = torch.from_numpy(out["probs"]) if npol else None
            vals = out["value"] if want_v else None
        else:
            x = torch.from_numpy(f.astype(np.float32)).to(self.device)
            probs = nnet.SOFT(self.policy_net(x[:npol])).cpu() if npol else None
            vals = self.value_net(x).reshape(-1).cpu().numpy() if want_v else None
        if npol:
            # Categorical re-normalises (nnet.py:274); puct reads the entries as python floats
            probs = probs / probs.sum(-1, keepdim=True)
            for i in range(npol):
                nodes[i]._prior = probs[i].tolist()
                nodes[i]._dist = None
        if want_v:
            for i, n in enumerate(nodes):
                if n._value is None:
                    n._value = float(vals[i])


class _StatView:
    """tree.N / tree.V / tree.Q: mapping node -> statistic with defaultdict semantics (mcts.py:50-52)."""

    def __init__(self, tree, attr, zero):
        self._tree, self._attr, self._zero = tree, attr, zero

    def __getitem__(self, node):
        n = self._tree._table.get(node.key())
        return getattr(n, self._attr) if n is not None else self._zero

    def __setitem__(self, node, v):
        setattr(self._tree._intern(node), self._attr, v)

    def __contains__(self, node):
        return node.key() in self._tree._table

    def __len__(self):
        return len(self._tree._table)

    def __iter__(self):
        return iter(self._tree._table.values())

    def items(self):
        return [(n, getattr(n, self._attr)) for n in self._tree._table.values()]


class _ChildrenView:
    """tree.children: node -> set-like list of children, only for expanded nodes (mcts.py:53)."""

    def __init__(self, tree):
        self._tree = tree

    def _get(self, node):
        n = self._tree._table.get(node.key())
        return None if n is None else n._kids

    def __contains__(self, node):
        return self._get(node) is not None

    def __getitem__(self, node):
        k = self._get(node)
        if k is None:
            raise KeyError(node)
        return k

    def get(self, node, default=None):
        k = self._get(node)
        return default if k is None else k

    def __iter__(self):
        return (n for n in self._tree._table.values() if n._kids is not None)

    def __len__(self):
        return sum(1 for _ in self)


class MCTS:
    """Monte Carlo tree searcher (reference mcts.py:15-255) on a batched evaluator."""

    _dirichlet = dirichlet.Dirichlet(0.1 * torch.ones(go.N ** 2))

    def __init__(self, root, policy_net=None, value_net=None, **kwargs):
        if policy_net is None:
            raise TypeError("Missing required keywork argument: 'policy_net'")
        self.policy_net = policy_net
        self.value_net = value_net
        self.no_sim = kwargs.get("no_sim", True)
        if self.value_net is None and self.no_sim:
            raise TypeError("Keyword argument 'value_net' is required for no simulation mode")
        self.expand_thresh = kwargs.get("expand_thresh", 100)
        self.branch_num = kwargs.get("branch_num")
        self.exploration_weight = kwargs.get("exploration_weight", 4.0)
        self.noise_weight = kwargs.get("noise_weight", 0)
        if self.no_sim:
            self.value_net_weight = 1.0
        elif self.value_net is None:
            self.value_net_weight = 0.0
        else:
            self.value_net_weight = kwargs.get("value_net_weight", 0.5)
        self.device = kwargs.get("device", torch.device("cpu"))
        policy_net.to(self.device)
        if value_net is not None:
            value_net.to(self.device)
        # one evaluator may serve many trees (self-play); by default each tree makes its own
        self.evaluator = kwargs.get("evaluator") or Evaluator(policy_net, value_net, self.device,
                                                              kwargs.get("max_batch"))
        self.eager = kwargs.get("eager_children", True)
        self._table = {}
        self.N = _StatView(self, "N", 0)
        self.V = _StatView(self, "V", 0.0)
        self.Q = _StatView(self, "Q", 0)
        self.children = _ChildrenView(self)
        self.root = None
        if kwargs.get("defer_root", False):
            self._pending_root = root   # a driver will run set_root_gen() itself (lock-step pools)
        else:
            self.set_root(root)

    # ---- node table ------------------------------------------------------------------------------------
    def _intern(self, node):
        k = node.key()
        n = self._table.get(k)
        if n is None:
            self._table[k] = n = node
            node.tree = self
        return n

    def clear_cache(self):
        """Forget everything outside the current root's subtree."""
        self._prune()

    def _prune(self):
        keep, q = {}, [self.root]
        while q:
            n = q.pop()
            if n.key() in keep:
                continue
            keep[n.key()] = n
            if n._kids:
                q.extend(n._kids)
        self._table = keep

    # ---- evaluation plumbing --------------------------------------------------------------------------------
    def _ev(self):
        if self.evaluator is None:   # an unpickled tree: the caller puts the nets back by hand (mcts.py:105-107)
            if self.policy_net is None or (self.no_sim and self.value_net is None):
                raise TypeError("this tree was unpickled without its nets: set tree.policy_net / tree.value_net first")
            self.evaluator = Evaluator(self.policy_net, self.value_net, self.device)
        return self.evaluator

    def _eval_now(self, policy_nodes, value_nodes):
        self._ev().run([EvalRequest(policy_nodes, value_nodes)])

    def _drive(self, gen):
        try:
            req = next(gen)
            while True:
                self._ev().run([req])
                req = gen.send(None)
        except StopIteration as e:
            return e.value

    # ---- public surface ----------------------------------------------------------------------------------------
    def set_root(self, node):
        self._drive(self.set_root_gen(node))

    def set_root_gen(self, node):
        node = self._intern(node)
        self.root = node
        node.tree = self
        # the reference touches root.dist (one policy evaluation) and then expands the root
        # (mcts.py:153-157); here both go into one request: root policy + the children's values
        yield from self._expand_gen(node)
        if node._prior is None:
            yield EvalRequest([node], [])
        node._add_noise(self.noise_weight)

    def rollout(self, n=1, analyze_dict=None):
        """Do n rollouts from the root (mcts.py:133-151)."""
        self._drive(self.rollout_gen(n, analyze_dict))

    def rollout_gen(self, n=1, analyze_dict=None):
        for _ in range(n):
            path = yield from self._descend_gen()
            leaf = path[-1]
            if analyze_dict is not None and len(path) > 2:
                analyze_dict[path[1]] = path[1:]
            score = None if self.no_sim else self._simulate(leaf)
            if self.value_net is not None and leaf._value is None:
                yield EvalRequest([], [leaf])
            self._backpropagate(path, score, leaf._value)

    def choose(self, node=None):
        """Most visited child; choosing at the root re-roots the tree (mcts.py:110-131)."""
        if node is None:
            node = self.root
        node = self._table.get(node.key(), node)
        if node._terminal:
            return node
        if not node._kids:
            # not expanded, or no legal move at all (the reference's max() over an empty set raises
            # here): sample from the policy, passing as the last resort
            child = node.find_random_child()
            if node is self.root and node._kids is not None:
                self.set_root(child)
            return child
        best, best_n = None, None
        for c in node._kids:
            s = float("-inf") if c.N == 0 else c.N
            if best is None or s > best_n:
                best, best_n = c, s
        if node is self.root:
            self.set_root(best)
        return best

    def winrate(self, node=None):
        w = self.value_net_weight
        if node is None:
            node = self.root
        n = self._table.get(node.key())
        if n is not None and n.N > 0:
            v = ((1 - w) * n.Q + w * n.V) / n.N
            return (v + 1) / 2
        return 0

    # ---- the search ------------------------------------------------------------------------------------------------
    def _descend_gen(self):
        path = [self.root]
        node = self.root
        while True:
            if not node._kids:
                if node._kids is None and node.N > self.expand_thresh:
                    yield from self._expand_gen(node)
                return path
            node = self._puct_select(node)
            path.append(node)

    def _descend(self):
        return self._drive(self._descend_gen())

    def _expand(self, node):
        self._drive(self._expand_gen(self._intern(node)))

    def _expand_gen(self, node):
        if node._kids is not None:
            return
        if self.branch_num and node._prior is None:
            yield EvalRequest([node], [])
        fresh = node.find_children(k=self.branch_num) if self.branch_num else node.find_children()
        kids = [self._intern(c) for c in fresh]
        pol = [node] if node._prior is None else []
        val = [c for c in kids if c._value is None] if (self.eager and self.value_net is not None) else []
        node._kids = kids
        if pol or val:
            yield EvalRequest(pol, val)

    def _puct_select(self, node):
        kids = node._kids
        if node._prior is None:  # only reachable with eager_children=False
            self._eval_now([node], [])
        total = 0
        for c in kids:
            total += c.N
        if total == 0:
            total = 1
        sq, cw, w, prior = sqrt(total), self.exploration_weight, self.value_net_weight, node._prior
        best, best_s = None, None
        for c in kids:
            n = c.N
            avg = 0 if n == 0 else ((1 - w) * c.Q + w * c.V) / n
            s = -avg + (cw * prior[c.mv] * sq / (1 + n))
            if best is None or s > best_s:
                best, best_s = c, s
        return best

    def _backpropagate(self, path, reward, leaf_val):
        for node in reversed(path):
            node.N += 1
            if reward:
                node.Q += reward
                reward = -reward
            if self.value_net is not None:
                node.V += leaf_val
                leaf_val = -leaf_val

    def _simulate(self, node, gnu=False):
        """Policy playout to the end of the game scored by area (mcts.py:195-206; no gnugo)."""
        invert = node.turn % 2 != 0
        node = self._intern(node) if node.key() in self._table else node
        while not node._terminal:
            if node.tree is None:
                node.tree = self
            node = node.find_random_child()
        r = node.reward()
        return -r if invert else r

    # ---- pickling: the nets do not travel (mcts.py:93-108) ---------------------------------------------
    def __getstate__(self):
        d = self.__dict__.copy()
        for k in ("policy_net", "value_net", "evaluator"):
            d[k] = None
        return d

    def __setstate__(self, state):
        """mcts.py:97-108: restore the dicts, give every node its tree back, leave the nets to the caller
        (`tree.policy_net = ...; tree.value_net = ...`; the evaluator is rebuilt on first use)."""
        self.__dict__.update(state)
        self._relink()
        self.policy_net = self.value_net = self.evaluator = None

    def _relink(self):
        for n in self._table.values():
            n.tree = self
            for c in n._kids or ():
                c.tree = self
        for view in (self.N, self.V, self.Q, self.children):
            view._tree = self

    def __deepcopy__(self, memo):
        """mcts.py:81-90: an independent copy of the search state (root, N, V, Q, children) that shares the
        networks -- and here the evaluator -- with the original."""
        new = self.__class__.__new__(self.__class__)
        new.__dict__.update(self.__dict__)
        clones = {}

        def clone(n):
            c = clones.get(id(n))
            if c is None:
                clones[id(n)] = c = n._clone_with_stats()
            return c

        new._table = {k: clone(n) for k, n in self._table.items()}
        for n in list(self._table.values()):
            if n._kids is not None:
                clone(n)._kids = [clone(k) for k in n._kids]
        new.root = clone(self.root) if self.root is not None else None
        new.N, new.V, new.Q = _StatView(new, "N", 0), _StatView(new, "V", 0.0), _StatView(new, "Q", 0)
        new.children = _ChildrenView(new)
        new._relink()
        return new
