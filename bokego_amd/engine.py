"""LeafEngine: Python handle of one bk_engine (one consumer thread, one MI355X).

Batches positions through the fused HIP kernel.  Mirrors what the reference does one
position at a time in nnet.policy_dist / nnet.value (bokego/nnet.py:265-284).
"""
import ctypes

import numpy as np

from . import _lib as L

_CONV = (0, 3, 6, 9, 12, 15, 18)
_BN = (1, 4, 7, 10, 13, 16, 19)


def _np32(v):
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(v), dtype=np.float32)


def _ptr(a):
    return a.ctypes.data_as(L.c_float_p)


def _fill_trunk(t, sd, keep):
    def get(name, size):
        if name not in sd:
            raise KeyError(f"state_dict is missing '{name}'")
        a = _np32(sd[name]).reshape(-1)
        if a.size != size:
            raise ValueError(f"'{name}' has {a.size} elements, expected {size}")
        keep.append(a)
        return _ptr(a)

    for l, (c, b) in enumerate(zip(_CONV, _BN)):
        t.conv_w[l] = get(f"conv.{c}.weight", 128 * (27 * 25 if l == 0 else 128 * 9))
        t.conv_b[l] = get(f"conv.{c}.bias", 128)
        t.bn_w[l] = get(f"conv.{b}.weight", 128)
        t.bn_b[l] = get(f"conv.{b}.bias", 128)
        t.bn_mean[l] = get(f"conv.{b}.running_mean", 128)
        t.bn_var[l] = get(f"conv.{b}.running_var", 128)
    t.head_w = get("conv.21.weight", 128)
    t.head_b = get("conv.21.bias", 81)


_HEAD_KEYS = (("bn_w", "bn.weight", 1), ("bn_b", "bn.bias", 1), ("bn_mean", "bn.running_mean", 1),
              ("bn_var", "bn.running_var", 1), ("lin1_w", "lin1.weight", 64 * 81), ("lin1_b", "lin1.bias", 64),
              ("lin_bn_w", "lin_bn.weight", 64), ("lin_bn_b", "lin_bn.bias", 64),
              ("lin_bn_mean", "lin_bn.running_mean", 64), ("lin_bn_var", "lin_bn.running_var", 64),
              ("lin2_w", "lin2.weight", 64), ("lin2_b", "lin2.bias", 1))


class LeafEngine:
    """policy_sd / value_sd: mappings with the reference state_dict names (torch tensors or arrays)."""

    def __init__(self, policy_sd=None, value_sd=None, device_id=0, max_batch=4096, precision=None, lib_path=None):
        """lib_path: another build of the library (tests load the fault-injection build, _lib.HOOKS_LIB_PATH)."""
        if policy_sd is None and value_sd is None:
            raise TypeError("LeafEngine needs policy and/or value weights")
        lib = L.load(lib_path)
        pw, vw, keep = self._weight_structs(policy_sd, value_sd)
        h = ctypes.c_void_p()
        rc = lib.bk_engine_create(ctypes.byref(pw) if pw is not None else None,
                                  ctypes.byref(vw) if vw is not None else None, int(device_id), int(max_batch),
                                  ctypes.byref(h))
        if rc != L.BK_OK:
            raise RuntimeError(f"bk_engine_create failed: {L.STATUS_NAMES.get(rc, rc)}: "
                               f"{lib.bk_last_error(None).decode()}")
        self._lib, self._h = lib, h
        self.device_id, self.max_batch = int(device_id), int(max_batch)
        self.has_policy, self.has_value = policy_sd is not None, value_sd is not None
        self._pending = {}
        if precision is not None:
            self.set_precision(precision)

    @staticmethod
    def _weight_structs(policy_sd, value_sd):
        """state_dicts -> (bk_policy_weights | None, bk_value_weights | None, arrays to keep alive during the call)"""
        keep = []
        pw = vw = None
        if policy_sd is not None:
            pw = L.PolicyWeights()
            _fill_trunk(pw.trunk, policy_sd, keep)
        if value_sd is not None:
            vw = L.ValueWeights()
            _fill_trunk(vw.trunk, value_sd, keep)
            for field, name, size in _HEAD_KEYS:
                if name not in value_sd:
                    raise KeyError(f"state_dict is missing '{name}'")
                a = _np32(value_sd[name]).reshape(-1)
                if a.size != size:
                    raise ValueError(f"'{name}' has {a.size} elements, expected {size}")
                keep.append(a)
                setattr(vw.head, field, _ptr(a))
        return pw, vw, keep

    def set_weights(self, policy_sd=None, value_sd=None):
        """New weights into this engine (bk_engine_set_weights): what load_state_dict on a live net does.  A net left None
        keeps its weights; every ticket must have been waited for."""
        if self._pending:
            raise RuntimeError("tickets outstanding: wait() for them before replacing the weights")
        pw, vw, keep = self._weight_structs(policy_sd, value_sd)
        self._check(self._lib.bk_engine_set_weights(self._h, ctypes.byref(pw) if pw is not None else None,
                                                    ctypes.byref(vw) if vw is not None else None))
        del keep

    def set_precision(self, precision):
        """'f32': exact fp32 MFMA; 'f16x2': fp16 hi/lo split operands, fp32 accumulation (faster)."""
        if precision not in L.PRECISIONS:
            raise ValueError(f"precision must be one of {list(L.PRECISIONS)}")
        self._check(self._lib.bk_engine_set_precision(self._h, L.PRECISIONS[precision]))

    @property
    def precision(self):
        return {v: k for k, v in L.PRECISIONS.items()}[self._lib.bk_engine_get_precision(self._h)]

    # -- helpers ---------------------------------------------------------------------------
    def _check(self, rc):
        if rc < 0:
            msg = self._lib.bk_last_error(self._h).decode()
            name = L.STATUS_NAMES.get(rc, str(rc))
            if rc in (-1, -4, -5):
                raise ValueError(f"{name}: {msg}")
            raise RuntimeError(f"{name}: {msg}")
        return rc

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bk_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _want(logits, probs, value):
        return (L.BK_WANT_LOGITS if logits else 0) | (L.BK_WANT_PROBS if probs else 0) | (L.BK_WANT_VALUE if value else 0)

    @staticmethod
    def _feats(feats):
        a = np.asarray(feats)
        if a.dtype == np.uint8 or a.dtype == np.int8:
            a = np.ascontiguousarray(a).view(np.uint8)
            dt = L.BK_FEATS_U8
        else:
            a = np.ascontiguousarray(a, dtype=np.float32)
            dt = L.BK_FEATS_F32
        if a.ndim == 3:
            a = a[None]
        if a.ndim != 4 or a.shape[1:] != (27, 9, 9):
            raise ValueError(f"features must be [B,27,9,9], got {a.shape}")
        return a, dt

    # -- host-buffer API -------------------------------------------------------------------
    def submit(self, feats, logits=False, probs=True, value=True, n_policy=None):
        """Asynchronous evaluation of host features; returns a ticket for wait().
        n_policy: run the PolicyNet only on the first n_policy positions (MCTS expansion batches);
        'logits'/'probs' then have n_policy rows, 'value' always B."""
        a, dt = self._feats(feats)
        B = a.shape[0]
        npol = B if n_policy is None else int(n_policy)
        want = self._want(logits, probs, value)
        out = {}
        if logits:
            out["logits"] = np.empty((npol, 81), np.float32)
        if probs:
            out["probs"] = np.empty((npol, 81), np.float32)
        if value:
            out["value"] = np.empty((B,), np.float32)
        p = lambda k: out[k].ctypes.data if k in out else None  # noqa: E731
        t = self._check(self._lib.bk_submit_prefix(self._h, a.ctypes.data, dt, B, npol, want, p("logits"),
                                                   p("probs"), p("value")))
        self._pending[t] = (a, out)
        return t

    @staticmethod
    def _records(positions):
        """positions: np.uint8 [B,192] (bk_pos records) or a ctypes array of go.Pos."""
        if isinstance(positions, np.ndarray):
            a = np.ascontiguousarray(positions, np.uint8)
            if a.ndim != 2 or a.shape[1] != 192:
                raise ValueError(f"position records must have shape [B,192], got {a.shape}")
            return a
        import ctypes
        n = ctypes.sizeof(positions)
        if n % 192:
            raise ValueError("position records must be 192 bytes each")
        return np.frombuffer(positions, np.uint8).reshape(-1, 192)

    def submit_positions(self, positions, logits=False, probs=True, value=True, n_policy=None):
        """As submit(), from position records (go.Pos / bk_pos, liberty cache refreshed by the host): the GPU
        encodes the 27 planes itself.  Same results as submit(features_u8) bit for bit."""
        a = self._records(positions)
        B = a.shape[0]
        npol = B if n_policy is None else int(n_policy)
        want = self._want(logits, probs, value)
        out = {}
        if logits:
            out["logits"] = np.empty((npol, 81), np.float32)
        if probs:
            out["probs"] = np.empty((npol, 81), np.float32)
        if value:
            out["value"] = np.empty((B,), np.float32)
        p = lambda k: out[k].ctypes.data if k in out else None  # noqa: E731
        t = self._check(self._lib.bk_submit_positions(self._h, a.ctypes.data, B, npol, want, p("logits"), p("probs"),
                                                      p("value")))
        self._pending[t] = (a, out)
        return t

    def encode_positions(self, positions):
        """The GPU encoder alone: uint8 [B,27,9,9] planes of the given position records."""
        a = self._records(positions)
        out = np.empty((a.shape[0], 27, 9, 9), np.uint8)
        self._check(self._lib.bk_encode_positions(self._h, a.ctypes.data, a.shape[0], out.ctypes.data))
        return out

    def wait(self, ticket):
        a, out = self._pending.pop(ticket)
        self._check(self._lib.bk_wait(self._h, ticket))
        return out

    def eval(self, feats, logits=False, probs=True, value=True, n_policy=None):
        """Synchronous: dict with the requested 'logits' [B,81], 'probs' [B,81], 'value' [B]."""
        return self.wait(self.submit(feats, logits=logits, probs=probs, value=value, n_policy=n_policy))

    # -- device-resident API (torch CUDA/HIP tensors) -------------------------------------
    def eval_device(self, d_feats, logits=False, probs=True, value=True, stream=None, n_policy=None):
        """d_feats: torch tensor on this engine's GPU, float32 or uint8 [B,27,9,9].
        Runs on torch's current stream (or `stream`); returns dict of torch tensors."""
        import torch

        if not d_feats.is_cuda or d_feats.device.index != self.device_id:
            raise ValueError("eval_device needs a tensor on the engine's GPU")
        x = d_feats.contiguous()
        if x.dtype == torch.uint8:
            dt = L.BK_FEATS_U8
        elif x.dtype == torch.float32:
            dt = L.BK_FEATS_F32
        else:
            raise ValueError("features must be float32 or uint8")
        if x.dim() == 3:
            x = x.unsqueeze(0)
        if x.dim() != 4 or tuple(x.shape[1:]) != (27, 9, 9):
            raise ValueError(f"features must be [B,27,9,9], got {tuple(x.shape)}")
        B = x.shape[0]
        npol = B if n_policy is None else int(n_policy)
        out = {}
        if logits:
            out["logits"] = torch.empty((npol, 81), dtype=torch.float32, device=x.device)
        if probs:
            out["probs"] = torch.empty((npol, 81), dtype=torch.float32, device=x.device)
        if value:
            out["value"] = torch.empty((B,), dtype=torch.float32, device=x.device)
        s = stream if stream is not None else torch.cuda.current_stream(x.device).cuda_stream
        p = lambda k: out[k].data_ptr() if k in out else None  # noqa: E731
        self._check(self._lib.bk_eval_device_prefix(self._h, x.data_ptr(), dt, B, npol,
                                                    self._want(logits, probs, value), p("logits"), p("probs"),
                                                    p("value"), ctypes.c_void_p(s)))
        out["_keepalive"] = x
        return out

    # -- misc ---------------------------------------------------------------------------------
    def set_option(self, name, value):
        """A diagnostic switch of this engine (bk_engine_set_option): 'coop', 'coop3', 'force_nb', 'no_split', 'no_direct',
        'no_head_part', 'copy_threads', 'encode_overlap', 'no_fuse_encode', 'direct_rows'.  The environment is read once, when the engine is created; a live
        engine is changed here.  Results never depend on a switch."""
        self._check(self._lib.bk_engine_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name):
        v = ctypes.c_int(0)
        self._check(self._lib.bk_engine_get_option(self._h, name.encode(), ctypes.byref(v)))
        return v.value

    def options(self, **kw):
        """with engine.options(coop=0): ... -- switches set for the block, put back afterwards"""
        import contextlib

        @contextlib.contextmanager
        def scope():
            old = {k: self.get_option(k) for k in kw}
            try:
                for k, v in kw.items():
                    self.set_option(k, v)
                yield self
            finally:
                for k, v in old.items():
                    self.set_option(k, v)
        return scope()

    def evaluator(self):
        """This engine as the evaluator of the native step loop (bk_engine_evaluator -> selfplay.Evaluator struct)."""
        from .selfplay import EvaluatorStruct
        ev = EvaluatorStruct()
        self._check(self._lib.bk_engine_evaluator(self._h, ctypes.byref(ev)))
        return ev

    def set_profiling(self, on=True):
        self._check(self._lib.bk_engine_set_profiling(self._h, int(on)))

    def synchronize(self):
        self._check(self._lib.bk_engine_synchronize(self._h))

    def stats(self):
        s = L.Stats()
        self._check(self._lib.bk_stats(self._h, ctypes.byref(s)))
        return {f: getattr(s, f) for f, _ in L.Stats._fields_}
