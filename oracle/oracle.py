"""oracle/oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front-end of oracle/libbkref.so (oracle/nnet_ref.c), the CPU restatement of the
reference's PolicyNet/ValueNet forward (reference bokego/nnet.py:19-113, 265-284).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Parity of the oracle itself is pinned against tests/golden/*.npz, which were produced by
importing the reference (tools/gen_golden.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_TRUNK_KEYS = []
for _l, (_c, _b) in enumerate(zip((0, 3, 6, 9, 12, 15, 18), (1, 4, 7, 10, 13, 16, 19))):
    _TRUNK_KEYS += [f"conv.{_c}.weight", f"conv.{_c}.bias", f"conv.{_b}.weight", f"conv.{_b}.bias",
                    f"conv.{_b}.running_mean", f"conv.{_b}.running_var"]
_TRUNK_KEYS += ["conv.21.weight", "conv.21.bias"]
_VALUE_KEYS = ["bn.weight", "bn.bias", "bn.running_mean", "bn.running_var", "lin1.weight", "lin1.bias",
               "lin_bn.weight", "lin_bn.bias", "lin_bn.running_mean", "lin_bn.running_var",
               "lin2.weight", "lin2.bias"]


def build(force=False):
    so = os.path.join(_HERE, "libbkref.so")
    src = os.path.join(_HERE, "nnet_ref.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libbkref.so"], stdout=subprocess.DEVNULL)
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        fp = ctypes.POINTER(ctypes.c_float)
        pp = ctypes.POINTER(fp)
        _LIB.bkref_policy_forward.argtypes = [pp, fp, ctypes.c_int, fp, fp, fp]
        _LIB.bkref_policy_forward.restype = None
        _LIB.bkref_value_forward.argtypes = [pp, fp, ctypes.c_int, fp, fp]
        _LIB.bkref_value_forward.restype = None
        _LIB.bkref_set_threads.argtypes = [ctypes.c_int]
        _LIB.bkref_set_threads.restype = ctypes.c_int
    return _LIB


def set_threads(n=0):
    """Set (n>0) / query the OpenMP thread count used by the oracle."""
    return _lib().bkref_set_threads(int(n))


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) if a is not None else None


class _Net:
    def __init__(self, tensors, keys):
        self._keep = [np.ascontiguousarray(np.asarray(tensors[k]), dtype=np.float32).reshape(-1) for k in keys]
        self._tab = (ctypes.POINTER(ctypes.c_float) * len(keys))(*[_fp(a) for a in self._keep])


class OraclePolicy(_Net):
    """tensors: mapping with the reference PolicyNet state_dict names."""

    def __init__(self, tensors):
        super().__init__(tensors, _TRUNK_KEYS)

    def __call__(self, x, want_probs=False, want_acts=False):
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 27, 9, 9)
        B = len(x)
        logits = np.empty((B, 81), np.float32)
        probs = np.empty((B, 81), np.float32) if want_probs else None
        acts = np.empty((B, 7, 128, 9, 9), np.float32) if want_acts else None
        _lib().bkref_policy_forward(self._tab, _fp(x), B, _fp(logits), _fp(probs), _fp(acts))
        out = (logits,)
        if want_probs:
            out += (probs,)
        if want_acts:
            out += (acts,)
        return out[0] if len(out) == 1 else out


class OracleValue(_Net):
    """tensors: mapping with the reference ValueNet state_dict names."""

    def __init__(self, tensors):
        super().__init__(tensors, _TRUNK_KEYS + _VALUE_KEYS)

    def __call__(self, x, want_acts=False):
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 27, 9, 9)
        B = len(x)
        values = np.empty((B,), np.float32)
        acts = np.empty((B, 7, 128, 9, 9), np.float32) if want_acts else None
        _lib().bkref_value_forward(self._tab, _fp(x), B, _fp(values), _fp(acts))
        return (values, acts) if want_acts else values
