"""ctypes binding of libbokego_amd.so (C ABI: include/bokego_amd.h).

There is no CPU fallback: if the library or a gfx950 device is missing, the calls raise.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BK_LIB_PATH") or os.path.join(_HERE, "libbokego_amd.so")  # BK_LIB_PATH: diagnostic builds

c_float_p = ctypes.POINTER(ctypes.c_float)

BK_OK = 0
BK_WANT_LOGITS, BK_WANT_PROBS, BK_WANT_VALUE = 1, 2, 4
BK_FEATS_F32, BK_FEATS_U8 = 0, 1
BK_MAX_INFLIGHT = 4
PRECISIONS = {"f32": 0, "f16x2": 1}
BK_ABI_VERSION = 7

STATUS_NAMES = {0: "BK_OK", -1: "BK_ERR_ARG", -2: "BK_ERR_HIP", -3: "BK_ERR_OOM", -4: "BK_ERR_BATCH",
                -5: "BK_ERR_NO_NET", -6: "BK_ERR_NO_GPU"}


class TrunkWeights(ctypes.Structure):
    _fields_ = [("conv_w", c_float_p * 7), ("conv_b", c_float_p * 7), ("bn_w", c_float_p * 7),
                ("bn_b", c_float_p * 7), ("bn_mean", c_float_p * 7), ("bn_var", c_float_p * 7),
                ("head_w", c_float_p), ("head_b", c_float_p)]


class ValueHeadWeights(ctypes.Structure):
    _fields_ = [(n, c_float_p) for n in ("bn_w", "bn_b", "bn_mean", "bn_var", "lin1_w", "lin1_b", "lin_bn_w",
                                         "lin_bn_b", "lin_bn_mean", "lin_bn_var", "lin2_w", "lin2_b")]


class PolicyWeights(ctypes.Structure):
    _fields_ = [("trunk", TrunkWeights)]


class ValueWeights(ctypes.Structure):
    _fields_ = [("trunk", TrunkWeights), ("head", ValueHeadWeights)]


class Stats(ctypes.Structure):
    _fields_ = [("evals", ctypes.c_uint64), ("batches", ctypes.c_uint64), ("max_batch_seen", ctypes.c_uint64),
                ("kernel_ms_sum", ctypes.c_double), ("kernel_ms_count", ctypes.c_uint64),
                ("last_kernel_ms", ctypes.c_double), ("f16_overflow_fallbacks", ctypes.c_uint64),
                ("f16_device_overflow", ctypes.c_uint64), ("positions_encoded", ctypes.c_uint64),
                ("split_launches", ctypes.c_uint64), ("coop_launches", ctypes.c_uint64),
                ("coop_fallbacks", ctypes.c_uint64), ("mean_batch", ctypes.c_double),
                ("queue_wait_ms_sum", ctypes.c_double), ("queue_wait_count", ctypes.c_uint64),
                ("host_wait_ms_sum", ctypes.c_double), ("failed_submissions", ctypes.c_uint64)]


# every symbol include/bokego_amd.h declares: (restype, argtypes)
_P = ctypes.c_void_p
SYMBOLS = {
    "bk_abi_version": (ctypes.c_int, []),
    "bk_device_count": (ctypes.c_int, []),
    "bk_engine_create": (ctypes.c_int, [ctypes.POINTER(PolicyWeights), ctypes.POINTER(ValueWeights), ctypes.c_int,
                                        ctypes.c_int, ctypes.POINTER(_P)]),
    "bk_engine_destroy": (ctypes.c_int, [_P]),
    "bk_engine_set_weights": (ctypes.c_int, [_P, _P, _P]),
    "bk_eval": (ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_int, _P, _P, _P]),
    "bk_eval_u8": (ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_int, _P, _P, _P]),
    "bk_eval_device": (ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P, _P, _P]),
    "bk_submit": (ctypes.c_int64, [_P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P, _P]),
    "bk_wait": (ctypes.c_int, [_P, ctypes.c_int64]),
    "bk_submit_prefix": (ctypes.c_int64, [_P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P, _P]),
    "bk_eval_device_prefix": (ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P,
                                             _P, _P]),
    "bk_submit_positions": (ctypes.c_int64, [_P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P, _P]),
    "bk_encode_positions": (ctypes.c_int, [_P, _P, ctypes.c_int, _P]),
    "bk_engine_set_precision": (ctypes.c_int, [_P, ctypes.c_int]),
    "bk_engine_get_precision": (ctypes.c_int, [_P]),
    "bk_engine_set_profiling": (ctypes.c_int, [_P, ctypes.c_int]),
    "bk_stats": (ctypes.c_int, [_P, ctypes.POINTER(Stats)]),
    "bk_engine_max_batch": (ctypes.c_int, [_P]),
    "bk_plan_query": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    "bk_plan_flops": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                                     ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]),
    "bk_engine_synchronize": (ctypes.c_int, [_P]),
    "bk_engine_set_option": (ctypes.c_int, [_P, ctypes.c_char_p, ctypes.c_int]),
    "bk_engine_get_option": (ctypes.c_int, [_P, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]),
    "bk_has_test_hooks": (ctypes.c_int, []),
    "bk_engine_evaluator": (ctypes.c_int, [_P, _P]),
    "bk_last_error": (ctypes.c_char_p, [_P]),
}

HOOKS_LIB_PATH = os.path.join(_HERE, "libbokego_amd_hooks.so")   # make hooks: the fault-injection build, for tests only

_libs = {}


def load(path=None):
    """Load the shared library (raises RuntimeError with build instructions if absent).  path: another build of it
    (tests/test_gpu_hooks.py loads the -DBK_TEST_HOOKS build beside the shipped one); each path is loaded once."""
    path = os.path.abspath(path or LIB_PATH)
    if path not in _libs:
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} not found: build it with `make -C bokego_amd/csrc` "
                "(or python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
        # PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.so.1.  Two HSA
        # runtimes in one process cannot both open the GPU, so make torch's copy the one the
        # dynamic linker binds our NEEDED libamdhip64.so.7 to (same SONAME) by loading it first.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = ctypes.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.bk_abi_version() != BK_ABI_VERSION and not os.environ.get("BK_LIB_ANY_ABI"):  # (tools/ab_bits.py: older builds)
            raise RuntimeError(f"{os.path.basename(path)} ABI version mismatch; rebuild")
        _libs[path] = lib
    return _libs[path]
