"""Host-side mirror of the reference's bokego/nnet.py inference surface, backed by the HIP engine.

Same names and signatures as the reference:
    features(game) -> f32[27,9,9]                         nnet.py:182-262
    policy_dist(policy, game, device, fts) -> Categorical  nnet.py:265-275
    value(v, game, device, fts) -> float                   nnet.py:277-284
    policy_sample(policy, game, device, fts) -> LongTensor nnet.py:286-297
and drop-in network objects HipPolicyNet / HipValueNet that satisfy everything the reference
uses of PolicyNet / ValueNet on the inference path (`__call__`, `.to`, `.eval`,
`.load_state_dict`, `.state_dict`, `.parameters`, ValueNet.load_policy_dict; boke.py:30-38,
mcts.py:54-76).  Training (`.train()`, autograd) is out of scope: these are inference objects.
"""
import ctypes
from collections import OrderedDict

import numpy as np
import torch
from torch.distributions.categorical import Categorical

from . import go
from .engine import LeafEngine, _BN, _CONV, _HEAD_KEYS

SOFT = torch.nn.Softmax(dim=1)

_TRUNK_NAMES = []
for _c, _b in zip(_CONV, _BN):
    _TRUNK_NAMES += [f"conv.{_c}.weight", f"conv.{_c}.bias", f"conv.{_b}.weight", f"conv.{_b}.bias",
                     f"conv.{_b}.running_mean", f"conv.{_b}.running_var"]
_TRUNK_NAMES += ["conv.21.weight", "conv.21.bias"]
_VALUE_NAMES = _TRUNK_NAMES + [n for _, n, _ in _HEAD_KEYS]


class _HipNet:
    _names = _TRUNK_NAMES
    _is_value = False

    def __init__(self, state_dict=None, device_id=0, max_batch=1024, precision=None):
        """precision: None = the engine default ('f32', the reference's arithmetic width; env BK_PRECISION
        overrides) or 'f16x2' (opt-in split-fp16 fast path, see include/bokego_amd.h)."""
        self.precision = precision
        self._sd = None
        self._engine = None
        self._shared = False
        self.device_id = int(device_id)
        self.max_batch = int(max_batch)
        self.training = False
        if state_dict is not None:
            self.load_state_dict(state_dict)

    # ---- torch.nn.Module surface used by the reference ---------------------------------------
    def load_state_dict(self, state_dict, strict=True):
        sd = OrderedDict()
        for n in self._names:
            if n not in state_dict:
                raise KeyError(f"missing key '{n}' in state_dict")
            v = state_dict[n]
            v = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
            sd[n] = np.array(v, dtype=np.float32)
        if strict:
            extra = [k for k in state_dict if k not in sd and not k.endswith("num_batches_tracked")]
            if extra:
                raise KeyError(f"unexpected keys in state_dict: {extra[:4]}")
        if self._engine is not None:
            # a live engine takes the new weights in place (bk_engine_set_weights) -- also when it is shared with the other net
            # (fuse()) or held by a running search's evaluator: everything that evaluates through it sees them from now on,
            # as every caller of a torch module sees an optimizer step or a load_state_dict (bin/selfplay.py:80-84,117-119).
            # Refused (RuntimeError, nothing changed) while the engine has tickets outstanding.
            self._engine.set_weights(**{"value_sd" if self._is_value else "policy_sd": sd})
        self._sd = sd
        return self

    def state_dict(self):
        self._need_weights()
        out = OrderedDict()
        for k, v in self._sd.items():
            out[k] = torch.from_numpy(v.copy())
            if k.endswith("running_var"):
                out[k[:-len("running_var")] + "num_batches_tracked"] = torch.tensor(0)
        return out

    def parameters(self):
        self._need_weights()
        return (torch.from_numpy(v.copy()) for k, v in self._sd.items() if "running_" not in k)

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("HipPolicyNet/HipValueNet are inference engines; training is out of scope")
        return self.eval()

    def to(self, device):
        """Accepted for API compatibility (mcts.py:74-76).  A cuda device picks the GPU; 'cpu' only
        says where inputs/outputs live -- the computation always runs on the MI355X."""
        d = torch.device(device)
        if d.type == "cuda" and d.index is not None and d.index != self.device_id:
            self.device_id = d.index
            self._drop_engine()
        return self

    def share_memory(self):
        return self

    # ---- engine plumbing --------------------------------------------------------------------------
    def _need_weights(self):
        if self._sd is None:
            raise RuntimeError(f"{type(self).__name__}: no weights loaded (call load_state_dict first)")

    def _drop_engine(self):
        if self._engine is not None and not self._shared:
            self._engine.close()
        self._engine, self._shared = None, False

    def engine(self):
        if self._engine is None:
            self._need_weights()
            kw = {"value_sd" if self._is_value else "policy_sd": self._sd}
            self._engine = LeafEngine(device_id=self.device_id, max_batch=self.max_batch, precision=self.precision, **kw)
        return self._engine

    def _run(self, x, **want):
        """x: torch tensor / ndarray [B,27,9,9] (or [27,9,9]) -> dict of arrays/tensors."""
        eng = self.engine()
        if isinstance(x, torch.Tensor) and x.is_cuda:
            return eng.eval_device(x, **want), x.device
        a = x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
        if a.ndim == 3:
            a = a[None]
        out = {}
        for i in range(0, len(a), eng.max_batch):  # the reference accepts any B; chunk to max_batch
            o = eng.eval(a[i:i + eng.max_batch], **want)
            for k, v in o.items():
                out.setdefault(k, []).append(v)
        return {k: torch.from_numpy(np.concatenate(v)) for k, v in out.items()}, None


class HipPolicyNet(_HipNet):
    """(B,27,9,9) -> (B,81) logits, PolicyNet.forward (nnet.py:54-57) on the MI355X."""

    def __call__(self, x):
        out, _ = self._run(x, logits=True, probs=False, value=False)
        return out["logits"]

    forward = __call__

    def probs(self, x):
        """softmax(logits) computed in the kernel's epilogue."""
        out, _ = self._run(x, logits=False, probs=True, value=False)
        return out["probs"]


class HipValueNet(_HipNet):
    """(B,27,9,9) -> (B,1) in (-1,1), ValueNet.forward (nnet.py:109-113) on the MI355X."""
    _names = _VALUE_NAMES
    _is_value = True

    def __call__(self, x):
        out, _ = self._run(x, logits=False, probs=False, value=True)
        return out["value"].reshape(-1, 1)

    forward = __call__

    def load_policy_dict(self, policy_dict):
        """Overlay a PolicyNet state_dict onto the trunk (nnet.py:103-107)."""
        self._need_weights()
        new = OrderedDict(self._sd)
        for k in _TRUNK_NAMES:
            if k in policy_dict:
                v = policy_dict[k]
                new[k] = np.array(v.detach().cpu().numpy() if hasattr(v, "detach") else v, dtype=np.float32)
        return self.load_state_dict(new)


def fuse(policy_net, value_net, max_batch=None, precision=None):
    """Put a HipPolicyNet and a HipValueNet on ONE engine so a batch needs one kernel launch.
    Returns the shared LeafEngine (also what the batched MCTS uses)."""
    if not isinstance(policy_net, HipPolicyNet) or not isinstance(value_net, HipValueNet):
        raise TypeError("fuse() needs a HipPolicyNet and a HipValueNet")
    if policy_net._engine is not None and policy_net._engine is value_net._engine:
        return policy_net._engine
    policy_net._need_weights()
    value_net._need_weights()
    mb = int(max_batch or max(policy_net.max_batch, value_net.max_batch))
    eng = LeafEngine(policy_net._sd, value_net._sd, device_id=policy_net.device_id, max_batch=mb,
                     precision=precision or policy_net.precision or value_net.precision)
    for n in (policy_net, value_net):
        n._drop_engine()
        n._engine, n._shared, n.max_batch, n.device_id = eng, True, mb, policy_net.device_id
    return eng


# ---- the reference's free functions --------------------------------------------------------------
def _as_game(game):
    if isinstance(game, go.Game):
        return game
    # a reference-style game object (board string, ko, last_move, turn [, _libs]): import its state,
    # including the liberty cache so history-dependent planes agree with the reference
    g = go.Game(board=game.board, ko=game.ko, last_move=game.last_move, turn=game.turn)
    libs = getattr(game, "_libs", None)
    if libs is not None:
        ctypes.memmove(g._pos.libs, bytes(bytearray(libs)), 81)
        g._pos.libs_valid = 1
    return g


def features(game):
    """go.Game -> float32 (27,9,9) tensor, the 27 planes of the reference (nnet.py:182-262)."""
    g = _as_game(game)
    f = g.features_u8()
    if g is not game and getattr(game, "_libs", None) is not None:
        game._libs[:] = bytes(g._pos.libs)  # keep the caller's cache in step, as get_liberties() would
    return torch.from_numpy(f.astype(np.float32))


def policy_dist(policy, game, device=torch.device("cpu"), fts=None):
    """Categorical distribution over the 81 points (nnet.py:265-275)."""
    if fts is None:
        fts = features(game)
    fts = fts.unsqueeze(0).to(device)
    probs = SOFT(policy(fts)).squeeze(0)
    return Categorical(probs)


def value(v, game, device=torch.device("cpu"), fts=None):
    """Value of the position for the side to move as a python float (nnet.py:277-284)."""
    if fts is None:
        fts = features(game)
    fts = fts.unsqueeze(0).to(device)
    return v(fts).item()


def policy_sample(policy, game, device=torch.device("cpu"), fts=None):
    """One move sampled from the policy (nnet.py:286-297); uses torch's global RNG like the reference."""
    if fts is None:
        fts = features(game)
    fts = fts.unsqueeze(0).to(device)
    probs = SOFT(policy(fts)).squeeze(0)
    return Categorical(probs).sample()
