"""Native RCCL all-reduce of the self-play statistics (libbkcomm.so, include/bokego_comm.h).

The Python drivers normally use torch.distributed (backend "nccl" is RCCL on ROCm) for the one collective
of the path; this module is the same step through the C ABI, as a host without torch would do it:

    comm = NativeComm.create(rank, world, device_id, "/tmp/bk_comm_id")   # rank 0 writes the id file
    totals = comm.allreduce_sum(local_stats)                              # np.float64 vector, <= 4096
"""
import ctypes
import os
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
COMM_LIB_PATH = os.environ.get("BK_COMM_LIB_PATH") or os.path.join(_HERE, "libbkcomm.so")
ID_BYTES = 128
_VP = ctypes.c_void_p
_T_IMPORT = time.time()   # a rank's own start, for telling a stale id file from this launch's

# every symbol include/bokego_comm.h declares
COMM_SYMBOLS = {
    "bk_comm_abi_version": (ctypes.c_int, []),
    "bk_comm_unique_id": (ctypes.c_int, [_VP]),
    "bk_comm_init": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _VP, ctypes.c_int, ctypes.POINTER(_VP)]),
    "bk_comm_allreduce_sum_f64": (ctypes.c_int, [_VP, _VP, ctypes.c_int]),
    "bk_comm_broadcast_f32": (ctypes.c_int, [_VP, _VP, ctypes.c_int64, ctypes.c_int]),
    "bk_comm_barrier": (ctypes.c_int, [_VP]),
    "bk_comm_rccl_version": (ctypes.c_int, []),
    "bk_comm_device_pci": (ctypes.c_int, [_VP, ctypes.c_char_p, ctypes.c_int]),
    "bk_comm_rank": (ctypes.c_int, [_VP]),
    "bk_comm_world": (ctypes.c_int, [_VP]),
    "bk_comm_destroy": (ctypes.c_int, [_VP]),
    "bk_comm_last_error": (ctypes.c_char_p, []),
}
_lib = None


def commlib():
    global _lib
    if _lib is None:
        if not os.path.exists(COMM_LIB_PATH):
            raise RuntimeError(f"{COMM_LIB_PATH} not found: build it with `make -C bokego_amd/csrc`")
        try:
            import torch  # noqa: F401  (same reason as in _lib.py: one HIP/RCCL runtime per process)
        except ImportError:
            pass
        lib = ctypes.CDLL(COMM_LIB_PATH)
        for name, (res, args) in COMM_SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


class NativeComm:
    def __init__(self, rank, world, device_id, unique_id):
        self._h = None                       # (first: __del__ runs close() also when anything below raises -- a missing libbkcomm.so)
        self._lib = commlib()
        if len(unique_id) != ID_BYTES:
            raise ValueError("unique_id must be 128 bytes")
        self._h = _VP()
        buf = (ctypes.c_uint8 * ID_BYTES).from_buffer_copy(bytes(unique_id))
        if self._lib.bk_comm_init(int(rank), int(world), buf, int(device_id), ctypes.byref(self._h)):
            raise RuntimeError("bk_comm_init: " + self._lib.bk_comm_last_error().decode())
        self.rank, self.world = rank, world

    @staticmethod
    def unique_id():
        buf = (ctypes.c_uint8 * ID_BYTES)()
        lib = commlib()
        if lib.bk_comm_unique_id(buf):
            raise RuntimeError("bk_comm_unique_id: " + lib.bk_comm_last_error().decode())
        return bytes(buf)

    @staticmethod
    def _job_tag(job):
        """16 bytes that every rank of ONE launch derives identically and another launch does not: the caller's
        `job`, else BK_COMM_JOB (bench.py's launcher exports a fresh nonce per launch), else what
        torch.distributed.run exports (run id + master address and port).  The last one repeats across reruns with
        the default port and no run id, which is why create() also looks at the file's age."""
        import hashlib
        if job is None:
            job = os.environ.get("BK_COMM_JOB") or "|".join(os.environ.get(k, "") for k in
                                                             ("TORCHELASTIC_RUN_ID", "MASTER_ADDR", "MASTER_PORT"))
        return hashlib.sha256(str(job).encode()).digest()[:16]

    @classmethod
    def create(cls, rank, world, device_id, id_path, timeout=120.0, job=None, stale_s=None):
        """Rendezvous through a file every rank can see: rank 0 writes [job tag | id], the others wait for a file
        carrying THEIR job tag (a file left behind by another job, or by a crashed run with another tag, is
        ignored); rank 0 removes a stale file before writing and its own file once every rank has joined.
        A file with the right tag that is older than this rank's own start by more than `stale_s` seconds
        (BK_COMM_STALE_S, default 600: ranks of one launch start within seconds of each other, but the first import of
        torch on a fresh box can take minutes) is a crashed launch's leftover and is ignored as well, and a fresh one
        must read the same twice 50 ms apart (rank 0 replaces a leftover as its first action)."""
        tag = cls._job_tag(job)
        if stale_s is None:
            stale_s = float(os.environ.get("BK_COMM_STALE_S", "600"))
        if rank == 0:
            try:
                os.unlink(id_path)
            except FileNotFoundError:
                pass
            uid = cls.unique_id()
            tmp = f"{id_path}.{os.getpid()}"
            with open(tmp, "wb") as f:
                f.write(tag + uid)
            os.replace(tmp, id_path)
        else:
            t0 = time.time()
            while True:
                def read():
                    try:
                        with open(id_path, "rb") as f:
                            return f.read(), os.fstat(f.fileno()).st_mtime
                    except FileNotFoundError:
                        return b"", 0.0
                blob, mtime = read()
                if len(blob) == 16 + ID_BYTES and blob[:16] == tag and mtime >= _T_IMPORT - stale_s:
                    time.sleep(0.05)
                    if read()[0] == blob:
                        break
                if time.time() - t0 > timeout:
                    raise RuntimeError(f"no communicator id for this job at {id_path} after {timeout}s")
                time.sleep(0.01)
            uid = blob[16:]
        comm = cls(rank, world, device_id, uid)   # collective: returns once all ranks have joined
        if rank == 0:
            try:
                os.unlink(id_path)
            except FileNotFoundError:
                pass
        return comm

    def allreduce_sum(self, vec):
        a = np.ascontiguousarray(vec, np.float64).copy()
        if self._lib.bk_comm_allreduce_sum_f64(self._h, a.ctypes.data, a.size):
            raise RuntimeError("bk_comm_allreduce_sum_f64: " + self._lib.bk_comm_last_error().decode())
        return a

    def barrier(self):
        """All ranks have arrived (a one-word all-reduce): in front of a timed collective, so that its time is not the wait."""
        if self._lib.bk_comm_barrier(self._h):
            raise RuntimeError("bk_comm_barrier: " + self._lib.bk_comm_last_error().decode())

    def rccl_version(self):
        return int(self._lib.bk_comm_rccl_version())

    def device_pci(self):
        buf = ctypes.create_string_buffer(32)
        if self._lib.bk_comm_device_pci(self._h, buf, 32):
            raise RuntimeError("bk_comm_device_pci: " + self._lib.bk_comm_last_error().decode())
        return buf.value.decode()

    def broadcast_f32(self, vec, root=0):
        """rank `root`'s float32 vector to every rank (in place on a contiguous array; returns it)."""
        a = np.ascontiguousarray(vec, np.float32)
        if self._lib.bk_comm_broadcast_f32(self._h, a.ctypes.data, a.size, int(root)):
            raise RuntimeError("bk_comm_broadcast_f32: " + self._lib.bk_comm_last_error().decode())
        return a

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bk_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


def _selftest():
    """python -m bokego_amd.comm: one rank of a world-N check of libbkcomm.so (RANK / WORLD_SIZE / LOCAL_RANK and
    BK_COMM_ID_PATH from the environment): all-reduce of a 173-double vector (the self-play statistics' length) and a broadcast; prints one JSON line."""
    import json
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dev = int(os.environ.get("BK_COMM_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    t0 = time.perf_counter()
    c = NativeComm.create(rank, world, dev, os.environ.get("BK_COMM_ID_PATH", "/tmp/bk_comm_id"))
    t_init = time.perf_counter() - t0
    v = (np.arange(173, dtype=np.float64) + 1) * (rank + 1)
    out = c.allreduce_sum(v)                      # first call: includes RCCL's lazy channel set-up
    t0 = time.perf_counter()
    for _ in range(10):
        out = c.allreduce_sum(v)
    t_ar = (time.perf_counter() - t0) / 10
    w = np.full(1_000_003, float(rank), np.float32)
    w = c.broadcast_f32(w, root=world - 1)
    ok = bool(np.array_equal(out, (np.arange(173) + 1) * (world * (world + 1) / 2)) and (w == world - 1).all())
    print(json.dumps({"rank": rank, "world": c.world, "ok": ok, "init_s": t_init, "allreduce_ms": t_ar * 1e3}), flush=True)
    c.close()
    return 0 if ok else 1


if __name__ == "__main__":
    raise SystemExit(_selftest())
