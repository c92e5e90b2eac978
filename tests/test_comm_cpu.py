"""The id-file rendezvous of bokego_amd.comm.NativeComm.create without RCCL: which files a non-root rank accepts."""
import os
import threading
import time

import pytest

from bokego_amd import comm


@pytest.fixture
def fake_comm(monkeypatch):
    joined = []
    monkeypatch.setattr(comm.NativeComm, "__init__", lambda self, rank, world, dev, uid: joined.append((rank, bytes(uid))))
    monkeypatch.setattr(comm.NativeComm, "unique_id", staticmethod(lambda: bytes(range(128))))
    monkeypatch.setattr(comm.NativeComm, "close", lambda self: None)
    return joined


def test_rank0_writes_and_a_peer_reads_the_same_id(tmp_path, fake_comm):
    path = str(tmp_path / "id")
    t = threading.Thread(target=lambda: comm.NativeComm.create(1, 2, 0, path, timeout=20, job="j1"))
    t.start()
    time.sleep(0.2)
    # rank 0 unlinks its file once the (fake) collective init returns, so hold it until the peer has read
    orig = os.unlink
    comm.NativeComm.create(0, 2, 0, path + ".other", job="j1")          # another path: nothing for the peer
    assert t.is_alive()
    tag = comm.NativeComm._job_tag("j1")
    with open(path + ".tmp", "wb") as f:
        f.write(tag + bytes(range(128)))
    os.replace(path + ".tmp", path)
    t.join(20)
    assert not t.is_alive() and (1, bytes(range(128))) in fake_comm and orig is os.unlink


def test_leftovers_of_other_or_crashed_launches_are_ignored(tmp_path, fake_comm):
    path = str(tmp_path / "id")
    tag = comm.NativeComm._job_tag("mine")
    with open(path, "wb") as f:                       # another job's file
        f.write(comm.NativeComm._job_tag("theirs") + bytes(128))
    with pytest.raises(RuntimeError, match="no communicator id"):
        comm.NativeComm.create(1, 2, 0, path, timeout=0.3, job="mine")
    with open(path, "wb") as f:                       # the right tag, but written long before this process started
        f.write(tag + bytes(128))
    old = time.time() - 3600
    os.utime(path, (old, old))
    with pytest.raises(RuntimeError, match="no communicator id"):
        comm.NativeComm.create(1, 2, 0, path, timeout=0.3, job="mine")
    comm.NativeComm.create(1, 2, 0, path, timeout=0.3, job="mine", stale_s=7200)   # ... unless the caller says that is fine
    os.utime(path, None)                              # a fresh file is taken
    comm.NativeComm.create(1, 2, 0, path, timeout=2, job="mine")
    assert [r for r, _ in fake_comm] == [1, 1]


def test_job_tag_sources(monkeypatch):
    monkeypatch.delenv("BK_COMM_JOB", raising=False)
    monkeypatch.setenv("MASTER_PORT", "29500")
    a = comm.NativeComm._job_tag(None)
    monkeypatch.setenv("BK_COMM_JOB", "nonce-1")
    b = comm.NativeComm._job_tag(None)
    assert a != b and b == comm.NativeComm._job_tag("nonce-1") and len(a) == 16


def test_a_communicator_whose_constructor_failed_has_a_quiet_destructor(capsys, monkeypatch):
    """NativeComm.__del__ -> close() read self._h before __init__ had set it -- a missing libbkcomm.so (or no
    GPU: bk_comm_init refuses) printed `AttributeError: 'NativeComm' object has no attribute '_h'` from the destructor."""
    import gc
    monkeypatch.setattr(comm, "_lib", None)
    monkeypatch.setattr(comm, "COMM_LIB_PATH", "/nonexistent/libbkcomm.so")
    with pytest.raises(RuntimeError, match="not found"):
        comm.NativeComm(0, 1, 0, bytes(128))
    gc.collect()
    assert "AttributeError" not in capsys.readouterr().err
    monkeypatch.undo()
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="bk_comm_init"):       # the library is there, the GPU is not
            comm.NativeComm(0, 1, 0, bytes(128))
        gc.collect()
        assert "AttributeError" not in capsys.readouterr().err
    c = comm.NativeComm.__new__(comm.NativeComm)
    c.close()                                                            # no handle: nothing to do, no error
