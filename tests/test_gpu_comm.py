"""-m gpu: the native RCCL all-reduce (libbkcomm.so) -- one-rank communicator on the box's GPU, and the value it
returns inside self_play's statistics path."""
import numpy as np
import pytest
import torch  # noqa: F401

from bokego_amd import comm

pytestmark = pytest.mark.gpu


def test_native_allreduce_world1(tmp_path):
    c = comm.NativeComm.create(0, 1, 0, str(tmp_path / "id"))
    v = np.arange(173, dtype=np.float64) * 1.5 - 7
    out = c.allreduce_sum(v)
    assert np.array_equal(out, v) and out is not v
    assert np.array_equal(c.allreduce_sum(np.zeros(0)), np.zeros(0))
    with pytest.raises(RuntimeError):
        c.allreduce_sum(np.zeros(5000))
    # comm ABI 2 (round 6): the barrier in front of the timed all-reduce, and what the first multi-GPU line says about its fabric
    c.barrier()
    assert c.rccl_version() > 20000 and len(c.device_pci()) >= 12 and c.device_pci().count(":") == 2
    c.close()


def test_bench_line_over_rccl_at_one_rank():
    """bench.py with torch.distributed's nccl backend (= RCCL) forced on at one rank: the same code the driver's N > 1 runs take --
    process group on the device, barrier, the statistics' all-reduce timed apart from the wait, RCCL's version and the rank's card
    in the line."""
    import json
    import os
    import subprocess
    import sys
    from conftest import REPO
    env = dict(os.environ, BK_BENCH_FORCE_DIST="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "1", "--sustain", "0", "--no-cpu-baseline",
                          "--no-live-pmc", "--selfplay-games", "32", "--no-f16x2"], capture_output=True, text=True, timeout=600, cwd=REPO, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][-1])
    assert d["collective_backend"] == "nccl" and d["collective_ranks_seen"] == 1
    assert d["rccl_version"] and d["rccl_version"][0].isdigit() and d["rank_devices"][0]["pci"].count(":") == 2
    f = d["selfplay"]["f32"]
    assert 0 < f["stats_allreduce_ms"] < 50 and len(f["allreduce_wait_ms_per_rank"]) == 1 and f["games"] == 32


def test_native_broadcast_world1_and_python_wrapper(tmp_path):
    """bk_comm_broadcast_f32 (RCCL ncclBroadcast through a device buffer) and selfplay.broadcast_weights on a
    one-rank communicator: 2 x 3.9 MB of weights make the round trip through the GPU unchanged."""
    import os
    from bokego_amd import selfplay
    from bokego_amd.bkw import load_bkw
    from conftest import GOLDEN
    c = comm.NativeComm.create(0, 1, 0, str(tmp_path / "id"))
    v = np.random.default_rng(1).standard_normal(100_003).astype(np.float32)
    assert np.array_equal(c.broadcast_f32(v.copy()), v)
    with pytest.raises(RuntimeError):
        c.broadcast_f32(v, root=1)
    pw, vw = load_bkw(os.path.join(GOLDEN, "policy_19.bkw")), load_bkw(os.path.join(GOLDEN, "value_synth.bkw"))
    bp, bv = selfplay.broadcast_weights(pw, vw, native_comm=c)
    assert list(bp) == list(pw) and list(bv) == list(vw)
    assert all(np.array_equal(bp[k], pw[k]) and bp[k].shape == pw[k].shape for k in pw)
    assert all(np.array_equal(bv[k], vw[k]) for k in vw)
    c.close()


def test_self_play_statistics_through_native_comm(tmp_path):
    import os
    from bokego_amd import selfplay
    from bokego_amd.bkw import load_bkw
    from bokego_amd.engine import LeafEngine
    from conftest import GOLDEN
    eng = LeafEngine(load_bkw(os.path.join(GOLDEN, "policy_19.bkw")), load_bkw(os.path.join(GOLDEN, "value_synth.bkw")), max_batch=2048)
    ev = selfplay.EngineEvaluator(eng)
    c = comm.NativeComm.create(0, 1, 0, str(tmp_path / "id"))
    kw = dict(n_games=6, rollouts=40, expand_thresh=10, max_turns=20, cap=2048)
    local, total = selfplay.self_play(ev, native_comm=c, **kw)
    _, plain = selfplay.self_play(ev, **kw)
    assert total == plain and total["games"] == 6 and local["allreduce_s"] > 0
    c.close()


def test_native_c_self_play_driver():
    """examples/bk_selfplay.c: config 4's loop from plain C over the four C ABIs (no Python in the loop);
    deterministic, and the SAME generation as the Python driver: both run bk_pools_run with the engine's evaluator and
    normalise priors with bk_normalise_rows, so every statistic of the reduced vector is equal."""
    import json
    import os
    import subprocess
    from bokego_amd import selfplay
    from bokego_amd.bkw import load_bkw
    from bokego_amd.engine import LeafEngine
    from conftest import GOLDEN, REPO
    exe = os.path.join(REPO, "examples", "bk_selfplay")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "bokego_amd", "csrc"), "examples"])
    cmd = [exe, os.path.join(GOLDEN, "policy_19.bkw"), os.path.join(GOLDEN, "value_synth.bkw"), "24", "100"]
    runs = [json.loads(subprocess.run(cmd, capture_output=True, text=True, timeout=300, check=True).stdout) for _ in range(2)]
    assert runs[0]["moves_checksum"] == runs[1]["moves_checksum"]
    assert runs[0]["games"] == 24 and 24 * 60 < runs[0]["plies"] <= 24 * 81
    eng = LeafEngine(load_bkw(os.path.join(GOLDEN, "policy_19.bkw")), load_bkw(os.path.join(GOLDEN, "value_synth.bkw")), max_batch=8192)
    _, total = selfplay.self_play(selfplay.EngineEvaluator(eng), n_games=24, rollouts=100, cap=8192)
    assert total["plies"] == runs[0]["plies"] and total["black_wins"] == runs[0]["black_wins"]
    assert total["value_evals"] == runs[0]["value_evals"] and total["policy_evals"] == runs[0]["policy_evals"]
    assert total["n_root_values"] == runs[0]["n_root_values"] == runs[0]["plies"]
    assert sum(total["root_visit_hist"]) == runs[0]["root_visit_hist_sum"] > 0
    assert abs(total["sum_root_value"] - runs[0]["sum_root_value"]) < 1e-8 and abs(total["sum_abs_root_value"] - runs[0]["sum_abs_root_value"]) < 1e-8


# ---- more than one rank: these need two GPUs and skip themselves on the one-GPU box ----------------
def _two_gpus():
    return torch.cuda.device_count() >= 2   # counting devices does not initialise the GPU on this image


def _spawn_ranks(argv, world, extra_env, timeout=300):
    """`world` fresh processes of `argv`, one per GPU, with the torch.distributed.run environment; -> their stdouts."""
    import os
    import socket
    import subprocess
    import sys
    from conftest import REPO
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PYTHONPATH=REPO, **extra_env)
        procs.append(subprocess.Popen([sys.executable] + argv, env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, e[-2000:]
        outs.append(o)
    return outs


@pytest.mark.skipif(not _two_gpus(), reason="needs two GPUs")
def test_native_comm_world2(tmp_path):
    """libbkcomm.so between two ranks on two GPUs: RCCL all-reduce + broadcast over xGMI, id through the file rendezvous."""
    import json
    outs = _spawn_ranks(["-m", "bokego_amd.comm"], 2, {"BK_COMM_ID_PATH": str(tmp_path / "id"), "BK_COMM_JOB": "t-native-2"})
    res = [json.loads(o.strip().splitlines()[-1]) for o in outs]
    assert [r["rank"] for r in res] == [0, 1] and all(r["ok"] and r["world"] == 2 for r in res)


def test_self_play_cli_two_ranks_on_one_card_over_gloo():
    """`python -m bokego_amd.selfplay` as two ranks (the torch.distributed.run environment) sharing the box's one GPU over
    gloo (BK_SELFPLAY_BACKEND / BK_SELFPLAY_DEVICE: RCCL refuses two ranks on one card): the sharding gid % world, the
    all-reduce of the statistics and the max-over-ranks time are the N-GPU run's; the reduced statistics equal one rank's."""
    import json
    import os
    from bokego_amd import selfplay
    from bokego_amd.bkw import load_bkw
    from bokego_amd.engine import LeafEngine
    from conftest import GOLDEN
    outs = _spawn_ranks(["-m", "bokego_amd.selfplay", "--games", "16", "--rollouts", "60", "--max-turns", "30"], 2,
                        {"BK_SELFPLAY_BACKEND": "gloo", "BK_SELFPLAY_DEVICE": "0"})
    two = json.loads(outs[0].strip().splitlines()[-1])
    assert not [ln for ln in outs[1].splitlines() if ln.lstrip().startswith("{")]     # rank 0 alone reports
    eng = LeafEngine(load_bkw(os.path.join(GOLDEN, "policy_19.bkw")), load_bkw(os.path.join(GOLDEN, "value_synth.bkw")), max_batch=8192)
    _, one = selfplay.self_play(selfplay.EngineEvaluator(eng), n_games=16, rollouts=60, max_turns=30, cap=8192)
    eng.close()
    assert two["n_gpus"] == 2 and two["games"] == 16 == one["games"] and two["plies"] == one["plies"]
    assert two["black_wins"] == one["black_wins"] and two["leaf_evals"] == one["value_evals"]


@pytest.mark.skipif(not _two_gpus(), reason="needs two GPUs")
def test_torch_distributed_nccl_world2_self_play_shards():
    """bokego_amd.selfplay as two ranks over torch.distributed's nccl (= RCCL) backend: the reduced statistics of a
    small generation equal the one-rank run's."""
    import json
    import os
    from bokego_amd import selfplay
    from bokego_amd.bkw import load_bkw
    from bokego_amd.engine import LeafEngine
    from conftest import GOLDEN
    outs = _spawn_ranks(["-m", "bokego_amd.selfplay", "--games", "16", "--rollouts", "60", "--max-turns", "30"], 2, {})
    two = json.loads(outs[0].strip().splitlines()[-1])
    eng = LeafEngine(load_bkw(os.path.join(GOLDEN, "policy_19.bkw")), load_bkw(os.path.join(GOLDEN, "value_synth.bkw")), max_batch=8192)
    _, one = selfplay.self_play(selfplay.EngineEvaluator(eng), n_games=16, rollouts=60, max_turns=30, cap=8192)
    eng.close()
    assert two["n_gpus"] == 2 and two["games"] == 16 == one["games"] and two["plies"] == one["plies"]
    assert two["black_wins"] == one["black_wins"] and two["leaf_evals"] == one["value_evals"]
