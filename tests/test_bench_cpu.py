"""bench.py's CPU legs (the reference-CPU baselines) run without a GPU: keeps them from rotting between GPU runs."""
import json
import os
import sys

import numpy as np

from bokego_amd.bkw import load_bkw

from conftest import GOLDEN, REPO

sys.path.insert(0, REPO)
import bench  # noqa: E402


def test_cpu_baseline_legs_on_a_small_sample():
    pw, vw = load_bkw(os.path.join(GOLDEN, "policy_19.bkw")), load_bkw(os.path.join(GOLDEN, "value_synth.bkw"))
    x, recs = bench.make_workload(96, 0)
    assert x.shape == (96, 27, 9, 9) and x.dtype == np.float32 and recs.shape == (96, 192)
    visible, quota, physical = bench.usable_cores()
    assert visible >= physical >= 1 and (quota is None or quota >= 1)
    # the oracle's own outputs stand in for the GPU's: the cross-check inside the leg must then report zero
    from oracle.oracle import OraclePolicy, OracleValue
    cpu = bench.cpu_baseline(pw, vw, x, OraclePolicy(pw)(x), OracleValue(vw)(x), budget_s=2.0)
    assert cpu["unit"] == "leaf-evals/s" and cpu["value"] > 50 and cpu["cores"] >= 1 and cpu["kind"] == "port"
    assert cpu["torch_cpu_leaf_evals_per_s"]["one_thread"]["B1"] > 10
    assert cpu["timed_output_vs_oracle_sample"]["max_abs_dlogit_vs_oracle"] == 0.0
    sp = bench.selfplay_cpu_baseline(2, rollouts=30)
    assert sp["games"] == 2 and sp["games_per_min"] > 0 and 40 <= sp["mean_plies"] <= 90


# ---- the launcher: `python3 bench.py --gpus N` starts its own ranks -------------------------------
def _run_bench(args, env=None, timeout=180):
    import subprocess
    e = dict(os.environ, **(env or {}))
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, capture_output=True, text=True, timeout=timeout,
                          cwd=REPO, env=e)


def test_cpulist_round_trip():
    assert bench.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert bench.format_cpulist([11, 0, 1, 2, 3, 8, 10]) == "0-3,8,10-11"
    assert bench.parse_cpulist("") == []


def test_launch_plan_splits_whole_cores_by_numa_node():
    """An 8-GPU node as the pool describes it: 2 sockets x 64 cores x 2 threads, four GPUs per socket, a 128-CPU quota."""
    node_cpus = {0: list(range(0, 64)) + list(range(128, 192)), 1: list(range(64, 128)) + list(range(192, 256))}
    plan = bench.launch_plan(8, allowed=range(256), gpu_nodes=[0, 0, 0, 0, 1, 1, 1, 1], node_cpus=node_cpus, quota=128,
                             core_of={c: c % 128 for c in range(256)})
    seen = set()
    for r in plan["ranks"]:
        cpus = bench.parse_cpulist(r["cpus"])
        assert len(cpus) == 32 and r["host_threads"] == 12 and r["device"] == r["rank"] == r["local_rank"]
        assert set(cpus) <= set(node_cpus[r["numa_node"]]) and r["numa_node"] == (0 if r["rank"] < 4 else 1)
        assert {c % 128 for c in cpus} == {c % 128 for c in cpus if c < 128}, "SMT siblings stay with one rank"
        assert not seen & set(cpus)
        seen |= set(cpus)
    assert len(seen) == 256
    # topology not visible (this container): an even split of what the process may run on, nothing lost, nothing shared
    flat = bench.launch_plan(3, allowed=range(8), gpu_nodes=[], node_cpus={}, quota=None, core_of={})
    assert [r["cpus"] for r in flat["ranks"]] == ["0-1", "2-4", "5-7"]
    # a box that shows only SOME of the GPUs' topology (the one-GPU lease: node 0 of device 0 only) gets the flat split for every
    # rank -- never a NUMA slice for one rank and a flat one overlapping it for the others
    mixed = bench.launch_plan(4, allowed=range(256), gpu_nodes=[0], node_cpus=node_cpus, quota=16, core_of={c: c % 128 for c in range(256)})
    got = [c for r in mixed["ranks"] for c in bench.parse_cpulist(r["cpus"])]
    assert len(got) == len(set(got)) == 256 and all(r["numa_node"] == -1 for r in mixed["ranks"])
    # one-GPU rehearsal: every rank opens the same device; the quota bounds the threads
    reh = bench.launch_plan(2, allowed=range(256), gpu_nodes=[1], node_cpus=node_cpus, quota=16, device=0,
                            core_of={c: c % 128 for c in range(256)})
    assert [r["device"] for r in reh["ranks"]] == [0, 0] and [r["host_threads"] for r in reh["ranks"]] == [7, 7]   # 8 CPUs' worth each, one left to the HIP runtime
    # eight ranks under a 16-CPU quota: 2 CPUs' worth each -> the pools run on the calling thread alone
    tight = bench.launch_plan(8, allowed=range(256), gpu_nodes=[0, 0, 0, 0, 1, 1, 1, 1], node_cpus=node_cpus, quota=16,
                              core_of={c: c % 128 for c in range(256)})
    assert [r["host_threads"] for r in tight["ranks"]] == [1] * 8
    assert all(r["numa_node"] == 1 for r in reh["ranks"])


def test_launch_plan_for_eight_gpus_on_a_small_cpu_share():
    """the driver's 8-GPU run may own far fewer CPUs than the node has.  2 NUMA nodes, four GPUs each, and
    (a) an affinity mask of 16 CPUs = 8 cores x 2 threads, 4 cores per node, 16-CPU quota: one whole core per rank, local to
    its GPU; (b) the same mask on ONE node only (the other node's GPUs have no local CPU): a flat split, 2 CPUs each; (c) 8 CPUs
    in all: one each.  Never a rank without a CPU, never a CPU handed out twice, one host thread per rank."""
    core_of = {c: c % 128 for c in range(256)}
    node_cpus = {0: list(range(0, 64)) + list(range(128, 192)), 1: list(range(64, 128)) + list(range(192, 256))}
    gpus = [0, 0, 0, 0, 1, 1, 1, 1]

    def check(plan, n_each):
        seen = set()
        for r in plan["ranks"]:
            cpus = bench.parse_cpulist(r["cpus"])
            assert len(cpus) == n_each == r["n_cpus"] and r["host_threads"] == 1
            assert not seen & set(cpus)
            seen |= set(cpus)
        return seen

    allowed = [0, 1, 2, 3, 128, 129, 130, 131, 64, 65, 66, 67, 192, 193, 194, 195]
    plan = bench.launch_plan(8, allowed=allowed, gpu_nodes=gpus, node_cpus=node_cpus, quota=16, core_of=core_of)
    assert check(plan, 2) == set(allowed)
    for r in plan["ranks"]:
        cpus = bench.parse_cpulist(r["cpus"])
        assert r["numa_node"] == (0 if r["rank"] < 4 else 1) and set(cpus) <= set(node_cpus[r["numa_node"]])
        assert len({core_of[c] for c in cpus}) == 1                       # the two threads of one core
    one_node = list(range(0, 8)) + list(range(128, 136))
    plan = bench.launch_plan(8, allowed=one_node, gpu_nodes=gpus, node_cpus=node_cpus, quota=16, core_of=core_of)
    assert check(plan, 2) == set(one_node) and all(r["numa_node"] == -1 for r in plan["ranks"])
    plan = bench.launch_plan(8, allowed=range(8), gpu_nodes=gpus, node_cpus=node_cpus, quota=8, core_of=core_of)
    assert check(plan, 1) == set(range(8))
    # fewer CPUs than ranks: the ranks share what there is (every rank still has somewhere to run)
    plan = bench.launch_plan(8, allowed=range(4), gpu_nodes=[], node_cpus={}, quota=4, core_of={})
    assert all(r["n_cpus"] >= 1 and r["host_threads"] == 1 for r in plan["ranks"])


def test_plan_flag_prints_the_plan_and_starts_nothing():
    out = _run_bench(["--gpus", "4", "--steps", "3", "--plan"])
    assert out.returncode == 0, out.stderr[-1000:]
    plan = json.loads(out.stdout)
    assert plan["world"] == 4 and len(plan["ranks"]) == 4 and plan["master_addr"] == "127.0.0.1" and plan["master_port"] > 0
    assert plan["argv"] == ["--gpus", "4", "--steps", "3"] and plan["backend"] == "nccl"
    every = [c for r in plan["ranks"] for c in bench.parse_cpulist(r["cpus"])]
    assert len(every) == len(set(every)) and set(every) <= set(os.sched_getaffinity(0))      # nothing shared, nothing foreign
    if not plan["gpu_numa_nodes"]:           # no GPU topology in sysfs (the build container): everything allowed is dealt out;
        assert sorted(every) == sorted(os.sched_getaffinity(0))   # with one, only the CPUs of the GPUs' NUMA nodes are


def test_launcher_starts_n_ranks_and_relays_one_line():
    """The launcher end to end without a GPU: N children rendezvous through the environment it gave them (gloo), each
    pinned to its own CPU slice; the parent prints rank 0's line and nothing else on stdout."""
    out = _run_bench(["--gpus", "3", "--steps", "7"], env={"BK_BENCH_LAUNCH_SELFTEST": "1"})
    assert out.returncode == 0, out.stderr[-1000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["collective_ranks_seen"] == 3 and d["sum_of_ranks_plus_1"] == 6.0 and d["steps"] == 7
    cpus = [set(bench.parse_cpulist(c)) for c in d["rank_cpus"]]
    if len(os.sched_getaffinity(0)) >= 3:
        assert not (cpus[0] & cpus[1]) and not (cpus[1] & cpus[2]) and set().union(*cpus) <= set(os.sched_getaffinity(0))
    # the exit protocol of the real run: the other ranks were gone and rank 0 had the host's CPUs back before it printed
    assert d["other_ranks_gone"] == [True, True]
    assert set(bench.parse_cpulist(d["rank0_cpus_for_the_cpu_legs"])) == set(os.sched_getaffinity(0))


def test_eight_ranks_rendezvous_leave_and_rank_0_finishes_alone():
    """the first 8-rank run will be the driver's.  Everything of it that needs no GPU, at 8 ranks: the plan,
    eight fresh children, the gloo rendezvous, a collective, the other seven leaving after the last collective and rank 0
    taking the host's CPUs back for the CPU legs, one line relayed."""
    out = _run_bench(["--gpus", "8", "--steps", "20", "--warmup", "5"], env={"BK_BENCH_LAUNCH_SELFTEST": "1"}, timeout=300)
    assert out.returncode == 0, out.stderr[-1000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["collective_ranks_seen"] == 8 and d["sum_of_ranks_plus_1"] == 36.0 and d["steps"] == 20
    assert d["other_ranks_gone"] == [True] * 7
    assert all(bench.parse_cpulist(c) for c in d["rank_cpus"])           # no rank without a CPU
    assert set(bench.parse_cpulist(d["rank0_cpus_for_the_cpu_legs"])) == set(os.sched_getaffinity(0))


def test_launcher_returns_nonzero_and_ends_the_others_when_a_rank_fails():
    import time
    t0 = time.time()
    out = _run_bench(["--gpus", "2"], env={"BK_BENCH_LAUNCH_SELFTEST": "fail:1", "BK_BENCH_LAUNCH_TIMEOUT": "60"})
    assert out.returncode == 1 and "rank 1 exited with 3" in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.strip()] and time.time() - t0 < 60


def test_counter_rows_fold_into_per_step_values():
    """bench.py's live PMC passes: rows of rocprofv3's counter_collection.csv -> per-step values.  A step of the timed call
    is two launches (3-board rounds + 2-board tail); a launch of another size (not in every step) is left out."""
    def rows(name, grid, n, ctr, value, ns):
        return [{"Kernel_Name": name, "Grid_Size": str(grid), "Counter_Name": ctr, "Counter_Value": str(value + i % 2),
                 "Start_Timestamp": "1000", "End_Timestamp": str(1000 + ns)} for i in range(n)]
    head, tail = "bk_leaf_eval_kernel<3, false>", "bk_leaf_eval_kernel<2, false>"
    tot = {}
    assert bench.fold_counter_rows(rows(head, 1310720, 8, "FETCH_SIZE", 49000, 7_500_000) + rows(tail, 131072, 8, "FETCH_SIZE", 37000, 530_000)
                                   + rows(head, 91648, 1, "FETCH_SIZE", 9999, 800_000), tot)
    by = tot.pop("_by_launch")                          # the same figures launch by launch (the 3-board rounds / the 2-board tail)
    assert sorted(c["FETCH_SIZE"] for c in by.values()) == [37000.5, 49000.5] and all("grid" in k for k in by)
    assert tot == {"FETCH_SIZE": 49000.5 + 37000.5, "_ns_FETCH_SIZE": 8_030_000.0}     # + the kernels' durations in this counter's pass
    assert bench.fold_counter_rows(rows(head, 1310720, 8, "GRBM_GUI_ACTIVE", 1.43e8, 7_500_000) + rows(tail, 131072, 8, "GRBM_GUI_ACTIVE", 1.0e7, 530_000), tot)
    assert tot["_ns"] == 8_030_000 and abs(tot["GRBM_GUI_ACTIVE"] - (1.43e8 + 1.0e7 + 1)) < 1e-3
    three = [r for c, v in (("SQ_WAVES", 20480), ("SQ_INSTS_VALU_MFMA_MOPS_F32", 2.1768e9), ("SQ_VALU_MFMA_BUSY_CYCLES", 1.74e10))
             for r in rows(head, 1310720, 4, c, v, 7_500_000)]
    assert bench.fold_counter_rows(three, tot) and abs(tot["SQ_WAVES"] - 20480.5) < 1e-9 and "SQ_VALU_MFMA_BUSY_CYCLES" in tot
    assert tot["_ns_SQ_VALU_MFMA_BUSY_CYCLES"] == 7_500_000        # the child's kernel time under the MFMA-counter pass
    assert not bench.fold_counter_rows([], {})
    assert bench.under_profiler() is False
    from bokego_amd import _lib
    if _lib.load().bk_device_count() == 0:
        assert bench.live_counters(4096, "f32") is None      # no GPU here: the passes fail, the caller falls back


def test_ranks_started_by_torchrun_pin_themselves_to_their_slice_of_the_plan():
    """Under torch.distributed.run nobody hands a rank its CPUs: every rank derives the same plan from sysfs and takes its own
    slice (ranks were not pinned).  Through the selftest hook: two ranks with torchrun's environment."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   BK_BENCH_LAUNCH_SELFTEST="1", TORCHELASTIC_RUN_ID="none")
        env.pop("BK_BENCH_CPUS", None)
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], env=env, cwd=REPO,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs[1][1][-500:]
    d = json.loads([ln for ln in outs[0][0].splitlines() if ln.startswith("{")][-1])
    cpus = [set(bench.parse_cpulist(c)) for c in d["rank_cpus"]]
    allowed = set(os.sched_getaffinity(0))
    if len(allowed) >= 2:
        assert not (cpus[0] & cpus[1]) and cpus[0] | cpus[1] <= allowed and d["collective_ranks_seen"] == 2
