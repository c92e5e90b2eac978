"""BKW1: flat little-endian fp32 container for PolicyNet / ValueNet state dicts.

The reference stores checkpoints as torch pickles ``{"model_state_dict": ...}``
(reference boke.py:30-38, bin/selfplay.py:207-208).  The engine's C ABI takes
plain float pointers, so the on-disk form used by the C/C++ side (and by the
golden fixtures) is this trivially parseable container holding the *unfolded*
tensors under their reference state_dict names (``conv.0.weight`` ...).
BatchNorm folding happens inside the engine at create time, in fp64.

Layout::

    "BKW1" | u32 version(=1) | u32 n_tensors | u32 reserved
    n x { char name[48] (NUL padded) | u32 ndim | u32 dims[4] | u64 offset | u64 nelem }
    raw fp32 data (each tensor 16-byte aligned, offsets from file start)
"""
import struct
from collections import OrderedDict

import numpy as np

MAGIC = b"BKW1"
_ENTRY = struct.Struct("<48sI4IQQ")
_HEAD = struct.Struct("<4sIII")


def save_bkw(path, tensors):
    """tensors: mapping name -> array-like (converted to float32)."""
    items = [(k, np.ascontiguousarray(np.asarray(v), dtype="<f4")) for k, v in tensors.items()]
    off = _HEAD.size + _ENTRY.size * len(items)
    off = (off + 15) // 16 * 16
    entries, blobs = [], []
    for name, arr in items:
        if arr.ndim > 4:
            raise ValueError(f"{name}: ndim {arr.ndim} > 4")
        nb = name.encode("ascii")
        if len(nb) > 47:
            raise ValueError(f"tensor name too long: {name}")
        dims = list(arr.shape) + [0] * (4 - arr.ndim)
        entries.append(_ENTRY.pack(nb, arr.ndim, *dims, off, arr.size))
        blobs.append((off, arr.tobytes()))
        off = (off + arr.nbytes + 15) // 16 * 16
    with open(path, "wb") as f:
        f.write(_HEAD.pack(MAGIC, 1, len(items), 0))
        for e in entries:
            f.write(e)
        for o, b in blobs:
            f.seek(o)
            f.write(b)
        f.truncate(off)


def load_bkw(path):
    """Returns OrderedDict name -> float32 ndarray (0-d tensors come back 0-d)."""
    with open(path, "rb") as f:
        raw = f.read()
    magic, ver, n, _ = _HEAD.unpack_from(raw, 0)
    if magic != MAGIC or ver != 1:
        raise ValueError(f"{path}: not a BKW1 file")
    out = OrderedDict()
    for i in range(n):
        nb, ndim, d0, d1, d2, d3, off, nelem = _ENTRY.unpack_from(raw, _HEAD.size + i * _ENTRY.size)
        name = nb.rstrip(b"\0").decode("ascii")
        shape = (d0, d1, d2, d3)[:ndim]
        arr = np.frombuffer(raw, dtype="<f4", count=nelem, offset=off).reshape(shape).copy()
        out[name] = arr
    return out


def state_dict_to_tensors(sd):
    """torch state_dict -> plain fp32 arrays, dropping BN step counters."""
    out = OrderedDict()
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            continue
        out[k] = v.detach().cpu().numpy().astype(np.float32)
    return out


def convert_pt(pt_path, bkw_path):
    """``.pt`` checkpoint (reference format, boke.py:31-32) -> BKW1 file."""
    import torch

    ck = torch.load(pt_path, map_location="cpu")
    sd = ck["model_state_dict"] if "model_state_dict" in ck else ck
    save_bkw(bkw_path, state_dict_to_tensors(sd))


def tensors_to_state_dict(tensors):
    """BKW1 arrays -> torch state_dict (adds the BN counters torch expects)."""
    import torch

    sd = OrderedDict()
    for k, v in tensors.items():
        sd[k] = torch.from_numpy(np.array(v, dtype=np.float32))
        if k.endswith("running_var"):
            sd[k[: -len("running_var")] + "num_batches_tracked"] = torch.tensor(0, dtype=torch.int64)
    return sd
