#!/usr/bin/env python3
"""bench.py -- leaf-evals/s of the fused PolicyNet+ValueNet HIP engine on MI355X.

A "step" is one pass of the hot path over one batch: BASELINE.json configs[1], 4096 9x9 positions, policy
logits + softmax + value, inputs already resident in HBM.  With --gpus N every rank (one process per GPU) evaluates
its own 4096 positions per step: the path shards with no data-path collective ("weak" scaling).

Launching.  `python3 bench.py --gpus N` starts its own N ranks: the parent makes NO GPU call, picks a free port on
127.0.0.1, splits the host's CPUs between the ranks by the NUMA node of each rank's GPU, starts N fresh children
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment, the way the reference's
bin/selfplay.py:177-199 spawns one worker per core), relays rank 0's single JSON line and returns non-zero if any child
fails.  Under torch.distributed.run (WORLD_SIZE already set) the process is a rank; it derives the same plan and pins itself
to its slice (BK_BENCH_NO_PIN=1: off).  `--plan` prints the launch plan and exits without touching a GPU.  BK_BENCH_BACKEND=gloo BK_BENCH_DEVICE=0 rehearses the N-rank path on one card.

The headline (`value`, `dtype`, `roofline`) is the EXACT-fp32 kernel -- the reference's arithmetic width
(torch fp32, bokego/nnet.py:31-57,73-113) -- timed over exactly --steps launches between barriers.  Beside it:
  sustained     the same launches looped for >= 2 s (DVFS / power settle), reported next to the --steps figure
  parity        measured IN THIS RUN on the timed engine through the timed entry point: max |dlogit| / |dprob| /
                |dvalue| against the reference's recorded outputs for the 536 golden positions (the cpu_baseline leg
                also checks a sample of the timed workload output against the CPU oracle it has loaded anyway)
  f16x2         the opt-in split-fp16 variant, same measurements, priced against the f16 MFMA peak
  roofline      bound = MFMA: SURVEY 8d's 266,838,272 algorithmic FLOP per leaf-eval over the dense MFMA peak of the
                dtype the matrix unit executes; kernel time from HIP events on the launch stream; `traffic`, `mfma_busy`,
                `effective_clock_ghz` and the executed MFMA FLOP from rocprofv3 --pmc passes made IN THIS RUN over a child
                process that issues the same launch (N=1; --no-live-pmc or a missing rocprofv3: the committed summary, named)
  cpu_baseline  the reference's CPU path restated (oracle/torch_ref.py: the same torch ops on the host's cores, and
                oracle/nnet_ref.c, the plain-C port) on a bounded sample of the same workload
  selfplay      secondary, outside the timed region: BASELINE configs[3] (512 self-play games sharded over the
                ranks + the end-of-generation all-reduce), per precision, with a CPU baseline at N=1.
  small_batch_latency  secondary: kernel time of the batches a single-tree genmove issues (configs[2]/[4]: 1 policy row +
                ~60 value rows) and of a single position (configs[0]), one CU per board against the cooperative
                launch (4 resp. 12 CUs per board; bit-identical outputs); and of requests between the whole-board forms'
                ranges (150 / 300 boards: three boards shared by 4 / 2 CUs)
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FLOP_PER_LEAF = 266_838_272          # SURVEY 8d / BASELINE.md 3: valid taps, both nets
FLOP_POLICY_EVAL = 133_413_888       # ... of which the PolicyNet (66,706,944 MAC) ...
FLOP_VALUE_EVAL = 133_424_384        # ... and the ValueNet (66,712,192 MAC)
PEAK_FP32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md chip table (dense, fp32 in / fp32 acc)
PEAK_F16_MFMA_TFLOPS = 2500.0        # MI355X_MICROARCH.md chip table (dense f16/bf16 MFMA)
PEAK_HBM_GBPS = 8000.0               # MI355X_MICROARCH.md chip table (HBM3E)
BYTES_PER_LEAF = 8748 + 328          # compulsory HBM bytes (f32 planes in, 81 probs + value out)
BATCH = 4096
TOL = {"logit": 1e-4, "prob": 1e-5, "value": 1e-4}   # BASELINE.json north_star / SURVEY 8d config 1
KERNEL = {"f32": "bk_leaf_eval_kernel<3, false> (+ its <2, false> tail launch: one step = 10 rounds of 3-board workgroups "
                 "+ 1 of 2-board ones)",
          "f16x2": "bk_leaf_eval_f16_kernel<3>"}
DTYPE = {"f32": "f32", "f16x2": "f16x2 (fp16 hi/lo split operands = 22-bit significands, fp32 accumulate)"}


def make_workload(B, rank):
    """SURVEY 8d config-2 input: position i is a seeded random playout from the empty board
    (rng = default_rng(20260 + i), L = rng.integers(0, 61) uniformly random legal non-eye-filling
    moves) encoded by the build's own board engine in incremental mode -> f32 [B,27,9,9].  The recipe
    is pinned move-for-move against the reference's rules engine by tests/golden/playouts.json."""
    from bokego_amd.workload import make_batch
    return make_batch(B, seed_base=20260 + rank * B, dtype=np.float32, with_records=True)


def measured_counters(batch, precision):
    """What the newest committed rocprofv3 PMC summary of this kernel says (profiles/*_pmc_<precision>.json, produced by
    tools/profile_bench.sh + tools/summarize_prof.py): HBM-side bytes per launch, MFMA-pipe occupancy, effective clock and
    the FLOP the matrix unit executed per workgroup.  Read from a file, so it describes the commit that file was made at,
    not this run: every figure carries `source`."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", f"*_pmc_{precision}.json")))
    if not files:
        return None
    d = json.load(open(files[-1]))
    if d.get("batch") != batch:
        return None
    c = d.get("counters", {})
    out = {"source": f"from_file:profiles/{os.path.basename(files[-1])} (batch {d.get('batch')})",
           "hbm_traffic_bytes_per_launch": d.get("hbm_traffic_bytes_per_launch"),
           "mfma_busy": d.get("mfma_busy"), "effective_clock_ghz": d.get("effective_clock_ghz")}
    if out["mfma_busy"] is None and c.get("SQ_VALU_MFMA_BUSY_CYCLES") and c.get("GRBM_GUI_ACTIVE"):
        out["mfma_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (c["GRBM_GUI_ACTIVE"] / 8)      # 1,024 SIMDs, 8 XCDs
    if precision == "f32" and c.get("SQ_INSTS_VALU_MFMA_MOPS_F32") and c.get("SQ_WAVES"):
        # the counter ticks once per 512 FLOP; 8 waves per workgroup.  Per workgroup, so that launches of other sizes in
        # the profiled run (the parity check's 536 positions) do not dilute it
        out["executed_mfma_flop_per_workgroup"] = c["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512 / (c["SQ_WAVES"] / 8)
    return out


def executed_over_algorithmic(batch, n_cu=256):
    """fp32 kernel: FLOP the matrix unit executes / algorithmic FLOP for one launch of `batch` positions on both nets,
    from the tile tables the kernel is compiled from (C ABI bk_plan_flops: tiles x taps not skipped x k-steps x 2,048
    FLOP per v_mfma_f32_16x16x4_f32, over the engine's launch plan).  Also the 3-board workgroup's own figure."""
    import ctypes
    from bokego_amd import _lib
    lib = _lib.load()
    ex, al, nl = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
    if lib.bk_plan_flops(batch, batch, n_cu, 0, ctypes.byref(ex), ctypes.byref(al), ctypes.byref(nl)):
        return None
    wg3 = ctypes.c_double()
    lib.bk_plan_flops(3 * n_cu, 0, n_cu, 0, ctypes.byref(wg3), None, None)     # one full round of 3-board workgroups
    return {"ratio": ex.value / al.value, "executed_mfma_flop_per_launch": ex.value, "launches_per_step": nl.value,
            "executed_mfma_flop_per_3_board_workgroup": wg3.value / n_cu, "source": "bk_plan_flops (tile tables of the compiled kernel)"}


# ---- host CPU description -------------------------------------------------------------------------------------------
def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores():
    """(threads this process may run on, cgroup CPU quota or None, physical cores among them)."""
    visible = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        pass
    smt = 1
    try:
        sib = open("/sys/devices/system/cpu/cpu0/topology/thread_siblings_list").read().strip()
        smt = max(1, len(sib.replace("-", ",").split(",")))
    except OSError:
        pass
    physical = max(1, visible // smt)
    return visible, quota, physical


def cpu_baseline(pw, vw, x, gpu_logits=None, gpu_values=None, budget_s=20.0):
    """The reference's CPU path on this host, bounded to ~budget_s: (a) oracle/torch_ref.py -- the reference's own
    operators on torch CPU -- at B = 1 / 64 / 4096 with every usable physical core and with 1 thread (SURVEY 8d,
    BASELINE.md 4); (b) the plain-C port oracle/nnet_ref.c.  `value` = the fastest of them."""
    import torch
    from oracle.oracle import OraclePolicy, OracleValue, set_threads
    from oracle.torch_ref import TorchPolicy, TorchValue, leaf_eval

    visible, quota, physical = usable_cores()
    # a 1-GPU box gets a 16-CPU share of the host: more threads than that only fight over it
    cores = min(physical, quota) if quota else (16 if visible > 64 else physical)
    cores = int(os.environ.get("BK_CPU_THREADS", cores))
    P, V = TorchPolicy(pw), TorchValue(vw)
    xt = torch.from_numpy(x)
    old = torch.get_num_threads()
    t_all = time.perf_counter()

    def rate(B, reps, min_s):
        xb = xt[:B]
        leaf_eval(P, V, xb)
        n, t0 = 0, time.perf_counter()
        while n < reps or time.perf_counter() - t0 < min_s:
            leaf_eval(P, V, xb)
            n += 1
        return n * B / (time.perf_counter() - t0)

    res = {}
    for label, nthreads in (("all_cores", cores), ("one_thread", 1)):
        torch.set_num_threads(nthreads)
        res[label] = {"threads": nthreads, "B1": rate(1, 10, 0.5), "B64": rate(64, 2, 1.0 if nthreads > 1 else 0.5)}
        if nthreads > 1:
            res[label][f"B{len(x)}"] = rate(len(x), 1, min(4.0, budget_s / 4))
    torch.set_num_threads(old)
    # plain-C port: whole passes over the batch with the same number of threads, and one thread at B=64
    Pc, Vc = OraclePolicy(pw), OracleValue(vw)
    set_threads(1)
    t0 = time.perf_counter(); Pc(x[:64]); Vc(x[:64]); c_one = 64 / (time.perf_counter() - t0)
    set_threads(cores)
    Pc(x[:256]); Vc(x[:256])
    n = min(len(x), 2048)
    t0 = time.perf_counter(); Pc(x[:n]); Vc(x[:n]); c_all = n / (time.perf_counter() - t0)
    # while the oracle is at hand: a sample of the TIMED GPU output against it (the workload has no recorded reference outputs)
    check = None
    if gpu_logits is not None:
        sel = np.linspace(0, len(x) - 1, 48).astype(int)
        check = {"positions": len(sel), "max_abs_dlogit_vs_oracle": float(np.abs(gpu_logits[sel] - Pc(x[sel])).max()),
                 "max_abs_dvalue_vs_oracle": float(np.abs(gpu_values[sel] - Vc(x[sel])).max())}
        assert check["max_abs_dlogit_vs_oracle"] < 2 * TOL["logit"] and check["max_abs_dvalue_vs_oracle"] < TOL["value"], check
    torch_best = max(v for k, v in res["all_cores"].items() if k.startswith("B"))
    best = max(torch_best, c_all)
    return {"value": best, "unit": "leaf-evals/s", "cores": cores, "kind": "port",
            "which": "oracle/torch_ref.py (the reference's torch ops, oneDNN/MKL)" if torch_best >= c_all else "oracle/nnet_ref.c (plain C, OpenMP)",
            "sample": f"same workload batch: torch CPU at B=1/64/{len(x)} with {cores} threads and B=1/64 with 1 thread, "
                      f"C port {n} positions with {cores} threads and 64 with 1; {time.perf_counter() - t_all:.1f}s of CPU work",
            "cpu_model": cpu_model(), "host_cpus_visible": visible, "cgroup_cpu_quota": quota, "physical_cores_visible": physical,
            "torch_cpu_leaf_evals_per_s": res, "torch_threads_default": old, "timed_output_vs_oracle_sample": check,
            "c_port_leaf_evals_per_s": {"threads": cores, "all_cores": c_all, "one_thread_B64": c_one}}


def selfplay_cpu_baseline(cores, rollouts=400):
    """The reference's way of running configs[3] on the host: one sequential tree per process, one position per
    network call (oracle/mcts_ref.py + oracle/torch_ref.py, 1 thread each).  `cores` worker processes play ONE FULL
    GAME each, concurrently (fresh interpreters that never touch the GPU); games/min = games / slowest worker."""
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", PYTHONPATH=REPO)
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, "-m", "oracle.mcts_ref", "--seed", str(s), "--rollouts", str(rollouts)],
                              cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
             for s in range(cores)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
            outs.append(json.loads(o.strip().splitlines()[-1]))
        except Exception:  # noqa: BLE001  (a worker that fails only shrinks the sample)
            p.kill()
    wall = time.perf_counter() - t0
    if not outs:
        return None
    return {"games_per_min": len(outs) / wall * 60.0, "unit": "games/min", "cores": cores, "kind": "port",
            "games": len(outs), "wall_s": wall, "mean_plies": float(np.mean([o["plies"] for o in outs])),
            "mean_s_per_move_one_core": float(np.mean([o["seconds"] / max(1, o["plies"]) for o in outs])),
            "sample": f"{len(outs)} full games x {rollouts} rollouts/move, one sequential tree per process on one core each "
                      f"(batch-1 torch-CPU network calls, as the reference makes them), run concurrently"}


# ---- the launcher: `python3 bench.py --gpus N` starts its own N ranks (no GPU call in this process) -----------------
def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def format_cpulist(cpus):
    cpus, parts, i = sorted(set(cpus)), [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        parts.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(parts)


def gpu_numa_nodes(sysfs="/sys"):
    """NUMA node of every GPU in HIP's enumeration order, read from sysfs only (no HIP call: the launcher must not
    initialise the GPU): the KFD topology lists the GPUs (nodes with simd_count > 0) in the order HIP numbers them,
    `drm_render_minor` names each one's DRM device and that device's `numa_node` is the host node its PCIe root hangs
    off.  [] when the topology is not visible (containers without /sys/class/kfd)."""
    base = os.path.join(sysfs, "class/kfd/kfd/topology/nodes")
    nodes = []
    try:
        for n in sorted(os.listdir(base), key=lambda v: int(v) if v.isdigit() else 1 << 30):
            props = {}
            try:
                for line in open(os.path.join(base, n, "properties")):
                    k, _, v = line.strip().partition(" ")
                    props[k] = v
            except OSError:       # a GPU this container may not open: HIP does not number it either
                continue
            if int(props.get("simd_count", "0") or 0) <= 0:
                continue
            numa = -1
            try:
                numa = int(open(os.path.join(sysfs, f"class/drm/renderD{int(props['drm_render_minor'])}/device/numa_node")).read())
            except (OSError, KeyError, ValueError):
                pass
            nodes.append(numa)
    except (OSError, ValueError):
        return []
    return nodes


def numa_cpus(sysfs="/sys"):
    """{numa node: [cpus]} from sysfs ({} when not visible)."""
    base, out = os.path.join(sysfs, "devices/system/node"), {}
    try:
        for n in os.listdir(base):
            if n.startswith("node") and n[4:].isdigit():
                out[int(n[4:])] = parse_cpulist(open(os.path.join(base, n, "cpulist")).read())
    except OSError:
        pass
    return out


def smt_siblings(cpus, sysfs="/sys"):
    """{cpu: first cpu of its physical core} for the given cpus (identity when sysfs does not say)."""
    out = {}
    for c in cpus:
        try:
            out[c] = min(parse_cpulist(open(os.path.join(sysfs, f"devices/system/cpu/cpu{c}/topology/thread_siblings_list")).read()))
        except (OSError, ValueError):
            out[c] = c
    return out


def cpu_quota():
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        return None


def launch_plan(n, allowed=None, gpu_nodes=None, node_cpus=None, quota=None, device=None, threads_cap=16, core_of=None):
    """One entry per rank: the GPU it opens, the CPUs it is pinned to and the host threads it may use.
    Ranks whose GPUs hang off the same NUMA node share that node's allowed CPUs in equal slices of whole physical
    cores (SMT siblings stay with one rank); a GPU whose node is unknown (or has no allowed CPU) takes a slice of an
    even split of everything allowed.  Host threads per rank
    = min(threads_cap, slice size, cgroup CPU quota // n): the quota, not the affinity mask, is what a container owns."""
    allowed = sorted(allowed if allowed is not None else os.sched_getaffinity(0))
    core_of = smt_siblings(allowed) if core_of is None else core_of
    by_core = lambda cpus: sorted(cpus, key=lambda c: (core_of.get(c, c), c))  # noqa: E731
    gpu_nodes = gpu_numa_nodes() if gpu_nodes is None else gpu_nodes
    node_cpus = numa_cpus() if node_cpus is None else node_cpus
    quota = cpu_quota() if quota is None else quota
    devs = [device if device is not None else r for r in range(n)]
    node_of = [gpu_nodes[d] if d < len(gpu_nodes) else -1 for d in devs]
    local = {nd: [c for c in node_cpus.get(nd, []) if c in set(allowed)] for nd in set(node_of)}
    # NUMA-local slices only when EVERY rank's GPU has a known node with enough CPUs for its ranks: a mixture of local and
    # flat slices would hand the same CPUs out twice
    if any(nd < 0 or len(local.get(nd) or []) < node_of.count(nd) for nd in node_of):
        node_of = [-1] * n
    ranks = []
    for r in range(n):
        nd = node_of[r]
        peers = [q for q in range(n) if node_of[q] == nd]
        pool = local.get(nd) or []
        if nd < 0 or len(pool) < len(peers):
            pool, peers, nd = allowed, list(range(n)), -1
        pool = by_core(pool)
        cores = sorted({core_of.get(c, c) for c in pool})
        k = peers.index(r)
        if len(cores) >= len(peers):       # whole cores per rank
            mine = set(cores[k * len(cores) // len(peers):(k + 1) * len(cores) // len(peers)])
            cpus = [c for c in pool if core_of.get(c, c) in mine]
        else:
            cpus = pool[k * len(pool) // len(peers):(k + 1) * len(pool) // len(peers)] or pool
        threads = max(1, min(threads_cap, len(cpus), (quota // n) if quota else len(cpus)))
        # the pools' workers spin, and beside them the HIP runtime keeps one thread busy (tools/cpu_use_probe.py: 12 host
        # threads = 12.7 CPUs): leave room for it and for Python -- 4 CPUs of a large share, 1 of a small one (a rank that
        # owns 2 CPUs' worth of a cgroup quota runs its pools on the calling thread alone)
        threads = max(1, threads - 4 if threads > 8 else threads - 1)
        ranks.append({"rank": r, "local_rank": r, "device": devs[r], "numa_node": nd, "cpus": format_cpulist(cpus),
                      "n_cpus": len(cpus), "host_threads": threads})
    return {"world": n, "ranks": ranks, "cpus_allowed": len(allowed), "cgroup_cpu_quota": quota,
            "gpu_numa_nodes": gpu_nodes, "numa_nodes_visible": sorted(node_cpus)}


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(args, argv):
    """Parent of a self-started N-rank run.  Starts N children of this same file, relays rank 0's stdout (the one JSON
    line), lets every child's stderr through, and returns 0 only if every child returned 0.  When a child fails the
    others are ended by their exact PIDs (they would otherwise wait in a collective for ever)."""
    import selectors
    backend = os.environ.get("BK_BENCH_BACKEND", "nccl")
    device = int(os.environ["BK_BENCH_DEVICE"]) if "BK_BENCH_DEVICE" in os.environ else None
    plan = launch_plan(args.gpus, device=device)
    plan.update(master_addr="127.0.0.1", master_port=free_port(), backend=backend, job=os.urandom(8).hex(),
                argv=[a for a in argv if a != "--plan"])
    if args.plan:
        print(json.dumps(plan), flush=True)
        return 0
    procs = []
    for r in plan["ranks"]:
        env = dict(os.environ, RANK=str(r["rank"]), LOCAL_RANK=str(r["local_rank"]), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR=plan["master_addr"], MASTER_PORT=str(plan["master_port"]),
                   BK_BENCH_CPUS=r["cpus"], BK_BENCH_HOST_THREADS=str(r["host_threads"]), BK_COMM_JOB=plan["job"],
                   OMP_NUM_THREADS=str(r["host_threads"]), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + plan["argv"], env=env, cwd=REPO,
                                      stdout=subprocess.PIPE if r["rank"] == 0 else sys.stderr, text=r["rank"] == 0 or None))
    deadline = time.time() + float(os.environ.get("BK_BENCH_LAUNCH_TIMEOUT", "1500"))
    sel = selectors.DefaultSelector()
    sel.register(procs[0].stdout, selectors.EVENT_READ)
    open_pipe, failed = True, None
    while True:
        if open_pipe and sel.select(timeout=0.5):
            line = procs[0].stdout.readline()
            if line:      # the JSON line goes to stdout; library chatter on rank 0's stdout (gloo prints there) to stderr
                dst = sys.stdout if line.lstrip().startswith("{") else sys.stderr
                dst.write(line)
                dst.flush()
            else:
                open_pipe = False
                sel.unregister(procs[0].stdout)
        elif not open_pipe:
            time.sleep(0.2)
        codes = [p.poll() for p in procs]
        bad = [(i, c) for i, c in enumerate(codes) if c not in (None, 0)]
        if bad and failed is None:
            failed = bad[0]
            print(f"bench.py launcher: rank {failed[0]} exited with {failed[1]}; ending the other ranks", file=sys.stderr)
            t_end = time.time() + 10
            while time.time() < t_end and any(p.poll() is None for p in procs):
                time.sleep(0.2)
            for p in procs:
                if p.poll() is None:
                    p.terminate()
        if failed is None and time.time() > deadline:
            failed = (-1, "timeout")
            print("bench.py launcher: timeout; ending the ranks", file=sys.stderr)
            for p in procs:
                if p.poll() is None:
                    p.terminate()
        if all(c is not None for c in codes) and not open_pipe:
            break
        if failed is not None and all(p.poll() is not None for p in procs):
            break
        if failed is not None and time.time() > deadline + 20:
            for p in procs:
                if p.poll() is None:
                    p.kill()
    for p in procs:
        p.wait()
    return 0 if failed is None and all(p.returncode == 0 for p in procs) else 1


_CPUS_BEFORE_PINNING = None


def leave_the_group(dist, rank, world, pids=None):
    """The last collective is behind us: barrier, tear the group down; every rank but 0 returns False (and exits); rank 0 waits
    until the others are gone, takes back the CPUs it had before it pinned itself to its slice, and returns True -- it then
    runs the CPU legs and the counter passes alone, with the host's cores and the GPUs to itself."""
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return False
    if world > 1:
        t_end = time.time() + 15
        while pids and time.time() < t_end:
            alive = []
            for p in pids:
                try:
                    os.kill(p, 0)
                    if open(f"/proc/{p}/stat").read().rsplit(")", 1)[1].split()[0] != "Z":   # (a zombie waiting for its launcher is gone)
                        alive.append(p)
                except (OSError, IndexError):
                    pass
            pids = alive
            time.sleep(0.1)
        if not pids:
            time.sleep(0.5)
        if _CPUS_BEFORE_PINNING:
            try:
                os.sched_setaffinity(0, _CPUS_BEFORE_PINNING)
            except OSError:
                pass
    return True


def launch_selftest(args):
    """What a rank does under BK_BENCH_LAUNCH_SELFTEST (tests/test_bench_cpu.py: the launcher without a GPU): join a
    gloo group through the launcher's environment, all-reduce, rank 0 prints one line.  `fail:R` makes rank R exit 3
    before the rendezvous, `hang:R` makes it sleep, so the parent's clean-up of the surviving ranks is under test too."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    mode, _, who = os.environ["BK_BENCH_LAUNCH_SELFTEST"].partition(":")
    if mode == "fail" and rank == int(who):
        return 3
    dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t)
    cpus, pids = [None] * world, [None] * world
    dist.all_gather_object(cpus, format_cpulist(os.sched_getaffinity(0)))
    dist.all_gather_object(pids, os.getpid())
    seen = dist.get_world_size()
    # the real run's exit protocol: the other ranks leave, rank 0 goes on alone (CPU legs, counter passes) and prints the line
    if not leave_the_group(dist, rank, world, [p for p in pids if p != os.getpid()]):
        return 0
    gone = []
    for p in pids[1:]:
        try:
            os.kill(p, 0)
            gone.append(open(f"/proc/{p}/stat").read().rsplit(")", 1)[1].split()[0] == "Z")
        except (OSError, IndexError):
            gone.append(True)
    print(json.dumps({"n_gpus": world, "collective_ranks_seen": seen, "collective_backend": "gloo",
                      "sum_of_ranks_plus_1": float(t.item()), "rank_cpus": cpus, "steps": args.steps,
                      "host_threads": os.environ.get("BK_BENCH_HOST_THREADS"), "other_ranks_gone": gone,
                      "rank0_cpus_for_the_cpu_legs": format_cpulist(os.sched_getaffinity(0))}), flush=True)
    return 0


# ---- GPU measurements ---------------------------------------------------------------------------------------------------
def parity_in_run(eng, torch):
    """max |dlogit| / |dprob| / |dvalue| of THIS engine, through the timed entry point (eval_device), against the
    reference's recorded outputs for the 536 golden positions (tests/golden/nets.npz, made by tools/gen_golden.py)."""
    g = os.path.join(REPO, "tests", "golden")
    f = np.load(os.path.join(g, "features.npz"))["incremental"].astype(np.float32)
    n = np.load(os.path.join(g, "nets.npz"))
    o = eng.eval_device(torch.from_numpy(f).cuda(), logits=True, probs=True, value=True)
    torch.cuda.synchronize()
    d = {"logit": float(np.abs(o["logits"].cpu().numpy() - n["logits_b1"]).max()),
         "prob": float(np.abs(o["probs"].cpu().numpy() - n["probs_b1"]).max()),
         "value": float(np.abs(o["value"].cpu().numpy() - n["values_b1"]).max())}
    for k, v in d.items():
        if not v < TOL[k]:
            raise AssertionError(f"parity lost: max |d{k}| = {v:g} >= {TOL[k]:g} on the golden positions")
    return {"positions": len(f), "max_abs_dlogit": d["logit"], "max_abs_dprob": d["prob"], "max_abs_dvalue": d["value"],
            "tolerance": TOL, "against": "reference outputs recorded in tests/golden/nets.npz"}


def measure(eng, x, steps, warmup, barrier, reduce_max, sustain_s, torch):
    """Times exactly `steps` launches between barriers (the contract's figure), then the same launch looped for
    >= sustain_s seconds, then a per-launch spread.  Kernel durations come from HIP events on the launch stream."""
    B = x.shape[0]
    run = lambda: eng.eval_device(x, logits=True, probs=True, value=True)  # noqa: E731
    for _ in range(warmup):
        run()
    torch.cuda.synchronize()
    eng.set_profiling(True)
    s0 = eng.stats()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = run()
    barrier()
    dt_local = time.perf_counter() - t0
    s1 = eng.stats()
    dt = reduce_max(dt_local)
    kern_ms = (s1["kernel_ms_sum"] - s0["kernel_ms_sum"]) / max(1, s1["kernel_ms_count"] - s0["kernel_ms_count"])
    assert torch.isfinite(out["value"]).all() and torch.isfinite(out["probs"]).all() and torch.isfinite(out["logits"]).all()

    # sustained: keep the chip busy for >= sustain_s (20-launch chunks, one sync per chunk)
    sust = None
    if sustain_s > 0:
        s0 = eng.stats()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < sustain_s:
            for _ in range(20):
                run()
            torch.cuda.synchronize()
            n += 20
        t_s = time.perf_counter() - t0
        s1 = eng.stats()
        sust = {"seconds": t_s, "launches": n, "leaf_evals_per_s": n * B / t_s,
                "kernel_ms": (s1["kernel_ms_sum"] - s0["kernel_ms_sum"]) / max(1, s1["kernel_ms_count"] - s0["kernel_ms_count"])}
    each = []
    for _ in range(30):
        run()
        torch.cuda.synchronize()
        each.append(eng.stats()["last_kernel_ms"])
    eng.set_profiling(False)
    return {"dt": dt, "dt_local": dt_local, "kernel_ms": kern_ms, "sustained": sust, "out": out,
            "p10_p50_p90": [float(np.percentile(each, q)) for q in (10, 50, 90)]}


PMC_PASSES = (("FETCH_SIZE",), ("WRITE_SIZE",), ("GRBM_GUI_ACTIVE",),
              ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_WAVES"))


def fold_counter_rows(rows, tot):
    """One rocprofv3 --pmc pass (rows of *_counter_collection.csv, already filtered to the leaf kernel) -> per-STEP counter
    values added into `tot`: a step is every launch the timed call makes (3-board rounds + 2-board tail = two kernels, or two
    grids of one); launches that do not occur in every step are left out.  Per launch kind the mean over its dispatches, summed
    over the kinds; "_ns" accumulates the kernels' durations alongside GRBM_GUI_ACTIVE (for the clock), "_ns_<counter>" the same
    for every counter's own pass (the child's kernel time under that pass).  False: nothing usable."""
    kinds = {}
    for x in rows:
        kinds.setdefault((x["Kernel_Name"], x["Grid_Size"]), []).append(x)
    n = max((len(v) for v in kinds.values()), default=0)
    if n == 0:
        return False
    for v in kinds.values():
        if len(v) < n:                               # a launch that is not part of every step
            continue
        per = {}
        for x in v:
            per.setdefault(x["Counter_Name"], []).append((float(x["Counter_Value"]), int(x["End_Timestamp"]) - int(x["Start_Timestamp"])))
        import re
        m = re.search(r"bk_\w+(<[^>]*>)?", v[0]["Kernel_Name"])
        kind = (m.group(0) if m else v[0]["Kernel_Name"][:48]) + f" grid {v[0]['Grid_Size']}"
        for c, vals in per.items():
            tot.setdefault("_by_launch", {}).setdefault(kind, {})[c] = sum(a for a, _ in vals) / len(vals)
            tot[c] = tot.get(c, 0.0) + sum(a for a, _ in vals) / len(vals)
            tot["_ns_" + c] = tot.get("_ns_" + c, 0.0) + sum(b for _, b in vals) / len(vals)   # the kernels' durations in THIS pass
            if c == "GRBM_GUI_ACTIVE":
                tot["_ns"] = tot.get("_ns", 0.0) + sum(b for _, b in vals) / len(vals)
    return True


def live_counters(batch, precision, timeout_s=150, device=0):
    """The hardware counters of the timed launch, measured in THIS run: one rocprofv3 --pmc pass per counter group (separate
    passes, as MI355X_MICROARCH.md's HBM section prescribes) around a child process -- `python3 bench.py --pmc-child`, started
    after the timed region -- that issues the same B-position launch a few times.  Per step (= every launch of the step: the
    3-board rounds and the 2-board tail) -> HBM-side bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 FETCH_SIZE
    correction), MFMA-pipe occupancy, effective clock, executed fp32-MFMA FLOP.  None when rocprofv3 is not there or a pass
    fails: the caller then falls back to the committed PMC summary and says so."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None or under_profiler():
        return None
    kern = "bk_leaf_eval_f16_kernel" if precision == "f16x2" else "bk_leaf_eval_kernel"
    tot, t0 = {}, time.perf_counter()
    for ctrs in PMC_PASSES:
        d = tempfile.mkdtemp(prefix="bk_pmc_", dir="/tmp")
        try:
            # (the child is not a rank: none of the parent's rendezvous / pinning variables)
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR",
                                                                    "MASTER_PORT", "BK_BENCH_CPUS", "BK_BENCH_HOST_THREADS", "BK_BENCH_PINNED_BY",
                                                                    "OMP_NUM_THREADS", "TORCHELASTIC_RUN_ID")}
            r = subprocess.run([exe, "--pmc", *ctrs, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
                                "--pmc-child", "--batch", str(batch), "--precision", precision, "--pmc-device", str(device)],
                               capture_output=True, text=True, timeout=timeout_s, cwd=REPO, env=dict(env, TMPDIR="/tmp"))
            files = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None
            rows = [x for f in files for x in csv.DictReader(open(f)) if kern in x["Kernel_Name"]]
            if not fold_counter_rows(rows, tot):
                return None
        except (subprocess.TimeoutExpired, OSError, KeyError, ValueError):
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out = {"source": f"measured in this run: rocprofv3 --pmc, {len(PMC_PASSES)} separate passes over a child process issuing the same "
                     f"launch ({time.perf_counter() - t0:.0f} s)", "counters_per_step": {k: v for k, v in tot.items() if not k.startswith("_")}}
    if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
        out["hbm_traffic_bytes_per_launch"] = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024
        # ... and launch by launch: the 2-board tail's bytes and every write are the same in every run; what moves from run to
        # run and box to box (177-179 MB on one box, 188 MB in the round-4 driver run) is the 3-board launch's read traffic --
        # how often each XCD's 4 MB L2 re-fetches its net's 3.9 MB weight stream during the ten rounds (tools/pmc_traffic.sh)
        out["hbm_traffic_bytes_by_launch"] = {k: (2 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024
                                              for k, c in tot.get("_by_launch", {}).items() if "FETCH_SIZE" in c or "WRITE_SIZE" in c}
    if tot.get("GRBM_GUI_ACTIVE") and tot.get("_ns"):
        out["effective_clock_ghz"] = tot["GRBM_GUI_ACTIVE"] / 8 / tot["_ns"]
        # the CHILD's own kernel time per step (profiler timestamps), in the clock pass and in the MFMA-counter pass: the
        # factors below multiply to the fraction AT THIS TIME, not at the timed loop's (which runs at its own clock)
        out["pmc_kernel_ms"] = tot["_ns"] / 1e6
        if tot.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            out["mfma_busy"] = tot["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (tot["GRBM_GUI_ACTIVE"] / 8)
            if tot.get("_ns_SQ_VALU_MFMA_BUSY_CYCLES"):
                out["pmc_kernel_ms_mfma_pass"] = tot["_ns_SQ_VALU_MFMA_BUSY_CYCLES"] / 1e6
    if tot.get("SQ_INSTS_VALU_MFMA_MOPS_F32"):
        out["executed_mfma_flop_per_launch"] = tot["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512
    return out


def under_profiler():
    """True when this process itself runs under rocprofv3 / rocprofiler (tools/profile_bench.sh): no profiler inside a profiler."""
    return any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB")) or \
        any(k.startswith("ROCPROF") for k in os.environ)


def pmc_child(args):
    """What the profiled child does: the timed launch, a few times, nothing else."""
    import torch
    from bokego_amd.bkw import load_bkw
    from bokego_amd.engine import LeafEngine
    g = os.path.join(REPO, "tests", "golden")
    torch.cuda.set_device(args.pmc_device)
    eng = LeafEngine(load_bkw(os.path.join(g, "policy_19.bkw")), load_bkw(os.path.join(g, "value_synth.bkw")), device_id=args.pmc_device,
                     max_batch=args.batch, precision=args.precision)
    x = torch.from_numpy(make_workload(args.batch, 0)[0]).cuda()
    for _ in range(8):
        eng.eval_device(x, logits=True, probs=True, value=True)
    torch.cuda.synchronize()
    eng.close()
    return 0


def roofline(precision, batch, kern_ms, sust, spread, live=None):
    f16 = precision == "f16x2"
    peak = PEAK_F16_MFMA_TFLOPS if f16 else PEAK_FP32_MFMA_TFLOPS
    achieved = batch * FLOP_PER_LEAF / (kern_ms * 1e-3) / 1e12
    pmc = dict(measured_counters(batch, precision) or {})
    file_source = pmc.get("source")
    if live:                                          # measured in this run: replaces what the committed summary says
        pmc.update({k: v for k, v in live.items() if v is not None})
        if live.get("executed_mfma_flop_per_launch") and precision == "f32":
            pmc.pop("executed_mfma_flop_per_workgroup", None)
    r = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
         "traffic": pmc.get("hbm_traffic_bytes_per_launch"), "traffic_by_launch": (live or {}).get("hbm_traffic_bytes_by_launch"),
         "traffic_source": pmc.get("source"), "kernel": KERNEL[precision],
         "kernel_ms": kern_ms, "kernel_ms_isolated_p10_p50_p90": spread,
         "algorithmic_flop_per_launch": batch * FLOP_PER_LEAF, "algorithmic_hbm_bytes_per_launch": batch * BYTES_PER_LEAF,
         "mfma_busy": pmc.get("mfma_busy"), "effective_clock_ghz": pmc.get("effective_clock_ghz"), "pmc_source": pmc.get("source"),
         "pmc_file_fallback": None if live else file_source}
    # the other roof, as BASELINE.json's north_star words it (HBM GB/s against the chip's peak): three orders of magnitude away
    r["hbm"] = {"peak_GBps": PEAK_HBM_GBPS, "algorithmic_GBps": batch * BYTES_PER_LEAF / (kern_ms * 1e-3) / 1e9,
                "measured_GBps": r["traffic"] / (kern_ms * 1e-3) / 1e9 if r["traffic"] else None}
    r["hbm"]["frac_measured"] = r["hbm"]["measured_GBps"] / PEAK_HBM_GBPS if r["hbm"]["measured_GBps"] else None
    r["hbm"]["note"] = ("measured = (2 x FETCH_SIZE + WRITE_SIZE) per step: the L2s' requests to the fabric, Infinity-Cache hits included "
                        "-- an upper bound on HBM bytes (both nets' workgroups read the planes; each XCD's L2 re-fetches its net's weights)")
    if live and live.get("executed_mfma_flop_per_launch"):
        r["pmc_executed_mfma_flop_per_launch"] = live["executed_mfma_flop_per_launch"]
    if live and live.get("pmc_kernel_ms"):
        # the counter child and the timed loop do not run at the same clock, so the line carries the
        # child's own kernel time and the fraction AT that time; mfma_busy x clock / 2.4 GHz / executed_ratio reproduces it
        r["pmc_kernel_ms"] = live["pmc_kernel_ms"]
        r["pmc_kernel_ms_mfma_pass"] = live.get("pmc_kernel_ms_mfma_pass")
        r["frac_at_pmc_kernel_ms"] = batch * FLOP_PER_LEAF / (live["pmc_kernel_ms"] * 1e-3) / 1e12 / peak
    if f16:
        # nominal: 3 half-precision MFMAs per algorithmic (fp32-equivalent) MAC, before padding and skipped zero-halo taps
        # (executed_mfma_tflops below is the counted figure for 3-board workgroups)
        r["executed_mfma_flop_per_algorithmic_flop"] = 3.0
        r["executed_ratio_source"] = "nominal (3 products per MAC)"
    else:
        ex = executed_over_algorithmic(batch)
        r["executed_mfma_flop_per_algorithmic_flop"] = ex["ratio"] if ex else None
        r["executed_ratio_source"] = ex["source"] if ex else None
        if ex:
            r["executed_mfma_flop_per_launch"] = ex["executed_mfma_flop_per_launch"]
            r["executed_mfma_tflops"] = ex["executed_mfma_flop_per_launch"] / (kern_ms * 1e-3) / 1e12
            r["executed_frac_of_peak"] = r["executed_mfma_tflops"] / peak
            if live and live.get("mfma_busy") and live.get("effective_clock_ghz"):
                r["frac_from_pmc_factors"] = live["mfma_busy"] * live["effective_clock_ghz"] / 2.4 / ex["ratio"]
                r["frac_from_pmc_factors_formula"] = "mfma_busy x effective_clock_ghz / 2.4 / executed_mfma_flop_per_algorithmic_flop = frac_at_pmc_kernel_ms"
            if pmc.get("executed_mfma_flop_per_workgroup"):     # the counters' view of the same number, per 3-board workgroup
                r["pmc_executed_mfma_flop_per_workgroup"] = pmc["executed_mfma_flop_per_workgroup"]
                r["tile_table_mfma_flop_per_3_board_workgroup"] = ex["executed_mfma_flop_per_3_board_workgroup"]
    if sust:
        r["sustained_kernel_ms"] = sust["kernel_ms"]
        r["sustained_frac"] = batch * FLOP_PER_LEAF / (sust["kernel_ms"] * 1e-3) / 1e12 / peak
    if f16 and batch >= 768:
        # per 3-board task and net 41,024 MFMAs of 32x32x16 (3x3 layers: 3 products, zero-halo taps skipped; layer 0: 2)
        r["executed_mfma_tflops"] = 2 * ((batch + 2) // 3) * 41024 * 32768 / (kern_ms * 1e-3) / 1e12
        r["frac_of_fp32_mfma_peak"] = achieved / PEAK_FP32_MFMA_TFLOPS
    return r


_T0 = time.time()


def progress(what):
    """one line per phase on stderr (rank 0 only): a long run shows where it is; BK_BENCH_QUIET=1: off"""
    if os.environ.get("RANK", "0") == "0" and not os.environ.get("BK_BENCH_QUIET"):
        print(f"bench.py [{time.time() - _T0:6.1f} s] {what}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--sustain", type=float, default=2.0, help="seconds of the sustained loop per variant (0: skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-selfplay", action="store_true", help="skip the secondary configs[3] measurement")
    ap.add_argument("--no-f16x2", action="store_true", help="skip the nested f16x2 block")
    ap.add_argument("--precision", choices=["f32", "f16x2"], default="f32",
                    help="arithmetic of the HEADLINE (default f32 = the reference's width; f16x2 only for profiling that variant)")
    ap.add_argument("--selfplay-games", type=int, default=512, help="games of the secondary configs[3] measurement (512 = the config; tests use fewer)")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not run the rocprofv3 --pmc passes (roofline.traffic etc. then come from the committed summary)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--pmc-device", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--plan", action="store_true", help="print the N-rank launch plan (devices, CPU slices, port) and exit; no GPU call")
    args = ap.parse_args()

    if args.plan or (args.gpus > 1 and "WORLD_SIZE" not in os.environ):
        sys.exit(launch(args, sys.argv[1:]))
    if "BK_BENCH_CPUS" not in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1 and not os.environ.get("BK_BENCH_NO_PIN"):
        # a rank started by torch.distributed.run: every rank derives the same plan from sysfs and takes its own slice
        try:
            dev = int(os.environ["BK_BENCH_DEVICE"]) if "BK_BENCH_DEVICE" in os.environ else None
            mine = launch_plan(int(os.environ["WORLD_SIZE"]), device=dev)["ranks"][int(os.environ.get("LOCAL_RANK", "0"))]
            os.environ["BK_BENCH_CPUS"], os.environ["BK_BENCH_HOST_THREADS"] = mine["cpus"], str(mine["host_threads"])
            os.environ["BK_BENCH_PINNED_BY"] = "rank"
        except (IndexError, ValueError, OSError):
            pass
    global _CPUS_BEFORE_PINNING
    _CPUS_BEFORE_PINNING = os.sched_getaffinity(0)
    if "BK_BENCH_CPUS" in os.environ:          # pin before torch starts its thread pools
        try:
            os.sched_setaffinity(0, parse_cpulist(os.environ["BK_BENCH_CPUS"]))
        except OSError as e:
            print(f"bench.py: could not pin rank to {os.environ['BK_BENCH_CPUS']}: {e}", file=sys.stderr)

    if os.environ.get("BK_BENCH_LAUNCH_SELFTEST"):
        sys.exit(launch_selftest(args))
    if args.pmc_child:
        sys.exit(pmc_child(args))

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the multi-rank path on a one-GPU box: BK_BENCH_BACKEND=gloo BK_BENCH_DEVICE=0 lets several ranks
    # share one card (RCCL refuses two ranks on one GPU); the numbers of such a run mean nothing
    backend = os.environ.get("BK_BENCH_BACKEND", "nccl")
    if "BK_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["BK_BENCH_DEVICE"])
    red_dev = "cuda" if backend == "nccl" else "cpu"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start {args.gpus} ranks (or unset WORLD_SIZE and let bench.py start them)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("BK_BENCH_FORCE_DIST"):  # the env var exercises the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from bokego_amd.bkw import load_bkw
    from bokego_amd.engine import LeafEngine

    g = os.path.join(REPO, "tests", "golden")
    pw, vw = load_bkw(os.path.join(g, "policy_19.bkw")), load_bkw(os.path.join(g, "value_synth.bkw"))
    eng = LeafEngine(pw, vw, device_id=local_rank, max_batch=max(args.batch, 8192), precision=args.precision)
    x_host, x_recs = make_workload(args.batch, rank)
    x = torch.from_numpy(x_host).cuda()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(v):
        if dist is None:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather(v):
        if dist is None:
            return [v]
        t = torch.tensor([v], dtype=torch.float64, device=red_dev)
        outl = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(outl, t)
        return [float(o.item()) for o in outl]

    # ---- headline: exactly --steps launches of the reference-width kernel, then the sustained loop -------------
    progress(f"engine up, workload made; timing {args.steps} steps")
    head = measure(eng, x, args.steps, args.warmup, barrier, reduce_max, args.sustain, torch)
    head_parity = parity_in_run(eng, torch)
    per_rank = gather(args.batch * args.steps / head["dt_local"])
    assert eng.stats()["f16_device_overflow"] == 0

    head_logits, head_values = head["out"]["logits"].cpu().numpy(), head["out"]["value"].cpu().numpy()   # for the CPU leg's cross-check

    # ---- nested: the opt-in f16x2 variant, same engine, same measurements --------------------------------------------
    other = None
    other_name = "f16x2" if args.precision == "f32" else "f32"
    if not args.no_f16x2:
        progress("headline timed; nested f16x2 block")
        eng.set_precision(other_name)
        m = measure(eng, x, args.steps, args.warmup, barrier, reduce_max, args.sustain, torch)
        other = {"value": world * args.batch * args.steps / m["dt"], "unit": "leaf-evals/s", "dtype": DTYPE[other_name],
                 "ms_per_step": m["dt"] / args.steps * 1e3,
                 "sustained": m["sustained"], "parity": parity_in_run(eng, torch),
                 "roofline": roofline(other_name, args.batch, m["kernel_ms"], m["sustained"], m["p10_p50_p90"]),
                 "f16_device_overflow_redos": eng.stats()["f16_device_overflow"],
                 "max_abs_dlogit_vs_headline_on_workload": float((m["out"]["logits"] - head["out"]["logits"]).abs().max().item())}
        eng.set_precision(args.precision)

    # PCIe-inclusive rates through the host-buffer ABI (reported beside, never as `value`):
    # (a) synchronous bk_eval with f32 planes, as the reference's host tensors would arrive;
    # (b) what the ABI is built for: uint8 planes, three tickets in flight (the engine runs H2D, kernels
    #     and D2H on three streams chained by events); (c) 192-byte position records, planes encoded on the GPU.
    e2e = e2e_u8 = e2e_pos = None
    if rank == 0:
        eng.eval(x_host, logits=False, probs=True, value=True)
        t1 = time.perf_counter()
        for _ in range(3):
            eng.eval(x_host, logits=False, probs=True, value=True)
        e2e = 3 * args.batch / (time.perf_counter() - t1)
        x_u8 = x_host.astype(np.uint8)
        ref = eng.eval(x_u8, logits=False, probs=True, value=True)
        got = eng.wait(eng.submit_positions(x_recs, logits=False, probs=True, value=True))
        assert np.array_equal(ref["probs"], got["probs"]) and np.array_equal(ref["value"], got["value"])

        def pipelined(submit, n_e2e=16):
            pend, t1 = [], time.perf_counter()
            for _ in range(n_e2e):                   # three tickets in flight: H2D / kernel / D2H streams overlap
                pend.append(submit())
                if len(pend) == 3:
                    eng.wait(pend.pop(0))
            while pend:
                eng.wait(pend.pop(0))
            return n_e2e * args.batch / (time.perf_counter() - t1)
        e2e_u8 = pipelined(lambda: eng.submit(x_u8, logits=False, probs=True, value=True))
        e2e_pos = pipelined(lambda: eng.submit_positions(x_recs, logits=False, probs=True, value=True))

    # The genmove regime (BASELINE configs[2]/[4]: one tree, 1600 rollouts/move): expansion batches of one policy row +
    # 40..70 value rows, latency-bound.  Kernel time of such a batch with one CU per board against the cooperative
    # launch (4 CUs per board here; 12 for the single position of configs[0]).
    small = None
    if rank == 0 and args.precision == "f32":
        def kernel_us(B, reps=20):
            xs = x_u8[:B]
            for _ in range(3):
                eng.eval(xs, probs=True, value=True, n_policy=1)
            eng.set_profiling(True)
            s0 = eng.stats()
            for _ in range(reps):
                eng.eval(xs, probs=True, value=True, n_policy=1)
            s1 = eng.stats()
            eng.set_profiling(False)
            return (1e3 * (s1["kernel_ms_sum"] - s0["kernel_ms_sum"]) / (s1["kernel_ms_count"] - s0["kernel_ms_count"]),
                    s1["coop_launches"] - s0["coop_launches"], s1["coop_fallbacks"] - s0["coop_fallbacks"])
        small = {"what": "kernel time (HIP events) of one host-path evaluation of B boards + 1 policy row, us"}
        for B in (1, 62):
            with eng.options(coop=0):
                one_cu = kernel_us(B)
            coop = kernel_us(B)
            small[f"B{B}"] = {"one_cu_per_board_us": one_cu[0], "cooperative_us": coop[0], "cooperative_launches": coop[1],
                              "fallbacks": coop[2]}
        # requests between the whole-board forms' ranges: groups of three boards shared by 4 / 2 CUs (bk_leaf_eval_coop3_kernel)
        for B, key in ((150, "three_boards_on_4_cus_us"), (300, "three_boards_on_2_cus_us")):
            with eng.options(coop=0):
                whole = kernel_us(B)
            coop = kernel_us(B)
            small[f"B{B}"] = {"whole_board_workgroups_us": whole[0], key: coop[0], "cooperative_launches": coop[1], "fallbacks": coop[2]}
        # 81..96 tasks: three boards on EIGHT CUs (round 5) against the 2-CUs-per-board form it replaces there
        with eng.options(coop3=0):
            two = kernel_us(90)
        eight = kernel_us(90)
        small["B90"] = {"two_cus_per_board_us": two[0], "three_boards_on_8_cus_us": eight[0], "cooperative_launches": eight[1], "fallbacks": eight[2]}
        # the remaining forms of the planner, so that every one has a time in the line and a row in the rocprofv3 summary of this command
        # (VERDICT r5 next #2a): 8 / 6 / 3 / 2 CUs per board
        for B, key in ((20, "eight_cus_per_board_us"), (33, "six_cus_per_board_us"), (70, "three_cus_per_board_us"), (110, "two_cus_per_board_us")):
            coop = kernel_us(B)
            small[f"B{B}"] = {key: coop[0], "cooperative_launches": coop[1], "fallbacks": coop[2]}
        # how far each of these requests is from the fp32-MFMA roof (B ValueNet rows, the first of which also runs the PolicyNet: B + 1 network tasks of 133.4 MFLOP)
        for k, v in small.items():
            if k.startswith("B"):
                tasks = int(k[1:]) + 1
                best = min(t for n, t in v.items() if n.endswith("_us"))
                v["tasks"] = tasks
                v["frac_of_fp32_mfma_peak"] = tasks * FLOP_PER_LEAF / 2 / (best * 1e-6) / 1e12 / PEAK_FP32_MFMA_TFLOPS

    # Secondary measurement (outside the timed region above): BASELINE configs[3] -- 512 self-play games,
    # 400 rollouts/move, sharded over the ranks (gid % world), one all-reduce of the statistics at the end.
    sp = None
    if not args.no_selfplay:
        from bokego_amd import selfplay
        # host cores are shared by the ranks: a launch() child got its slice and thread count from the plan, a
        # torch.distributed.run rank sees every CPU and takes 1/world of them (of the cgroup quota when there is one)
        threads = int(os.environ.get("BK_BENCH_HOST_THREADS", 0))
        if not threads:
            share = max(1, min(16, (cpu_quota() or len(os.sched_getaffinity(0))) // world))
            threads = max(1, share - 4 if share > 8 else share - 1)      # as launch_plan: room for the HIP runtime's thread
        host_threads = threads                                          # what this rank may use at all

        def threads_for(games_here):
            """<= one host thread per 8 games of a pool (two pools): selfplay.default_threads; the weak leg's 512 games per rank
            take more than the strong leg's 512 / n_gpus"""
            return max(1, min(12, host_threads, games_here // 16 or 1))
        threads = threads_for(args.selfplay_games // world)
        sp = {"config": f"configs[3]: {args.selfplay_games} games, 400 rollouts/move, games sharded gid % n_gpus",
              "collective": f"1 all-reduce of {selfplay.STATS_LEN} doubles per generation (scalars + first-move and root-visit histograms)",
              "host_threads_per_rank": threads, "step_loop": "C (bk_pools_run)" if selfplay.NATIVE_LOOP else "Python (run_pools)",
              "search": "an expansion evaluates its best-prior children only (bk_search_params.eager_top), the rest when a rollout "
                        "reaches it: the same 512 games as with every child evaluated (rounds 1-2), 3.4 M -> 0.6-0.7 M evaluations "
                        "(children_evaluated_per_expansion: fp32 2 from 192 games per rank, else 4; f16x2 6)"}
        red = torch.device("cuda", local_rank) if backend == "nccl" else None
        # untimed: a small generation first (the pools' worker threads exist, the allocator and the caches are warm)
        progress("self-play warm-up generation")
        selfplay.self_play(selfplay.EngineEvaluator(eng), n_games=min(64 * world, args.selfplay_games), rollouts=50, rank=rank, world=world,
                           cap=8192, threads=threads, reduce_device=red)

        def generation(n_games, prec, leaves=1):
            """one generation of n_games games over all ranks: whole-job games/min (max over ranks), what each rank needed, the
            all-reduce, and the visit / value statistics it sums"""
            ev = selfplay.EngineEvaluator(eng)
            barrier()
            mine = len(selfplay.shard_game_ids(n_games, rank, world))
            local, total = selfplay.self_play(ev, n_games=n_games, rollouts=400, rank=rank, world=world, cap=8192, threads=threads_for(mine),
                                              reduce_device=red, leaves=leaves)
            per_rank_s = gather(local["seconds"])
            reduce_ms = gather(local["allreduce_s"] * 1e3)          # the collective alone (every rank waited at a barrier first) ...
            wait_ms = gather(local["allreduce_wait_s"] * 1e3)       # ... and that wait: how long before the slowest rank this one was done
            secs = max(per_rank_s)
            # the leg on the roofline ("report games/min, leaf-evals/s/GPU"): network evaluations the generation asked
            # for x their algorithmic FLOP, over the leg's seconds -- whole job against n_gpus x the fp32-MFMA peak, and rank by rank
            i_val, i_pol = selfplay.STATS_FIELDS.index("value_evals"), selfplay.STATS_FIELDS.index("policy_evals")
            rank_val, rank_pol = gather(float(local["local_stats"][i_val])), gather(float(local["local_stats"][i_pol]))
            req, sent = gather(float(local["rows_requested"])), gather(float(local["rows_sent"]))
            flop = total["policy_evals"] * FLOP_POLICY_EVAL + total["value_evals"] * FLOP_VALUE_EVAL
            peak = PEAK_F16_MFMA_TFLOPS if prec == "f16x2" else PEAK_FP32_MFMA_TFLOPS
            rank_tf = [(p * FLOP_POLICY_EVAL + v * FLOP_VALUE_EVAL) / t / 1e12 if t > 0 else 0.0 for p, v, t in zip(rank_pol, rank_val, per_rank_s)]
            roof = {"bound": "mfma", "unit": "TFLOP/s", "policy_evals": int(total["policy_evals"]), "value_evals": int(total["value_evals"]),
                    "algorithmic_flop": float(flop), "flop_per_policy_eval": FLOP_POLICY_EVAL, "flop_per_value_eval": FLOP_VALUE_EVAL,
                    "achieved_tflops": flop / secs / 1e12, "peak": peak * world, "frac": flop / secs / 1e12 / (peak * world),
                    "per_rank_achieved_tflops": rank_tf, "per_rank_frac": [t / peak for t in rank_tf],
                    # equal rows of one batch travel once (in-batch de-duplication) while every asker counts its evaluation: what the
                    # GPU executed is the smaller number
                    "rows_sent_over_rows_requested": sum(sent) / sum(req) if sum(req) > 0 else 1.0,
                    "frac_of_rows_sent": flop / secs / 1e12 / (peak * world) * (sum(sent) / sum(req) if sum(req) > 0 else 1.0),
                    "what": "evaluations the searches asked for (reduced vector: policy_evals, value_evals) x algorithmic FLOP (valid taps) "
                            "/ the leg's seconds (max over ranks), end to end: host, launches and copies included"}
            return {"games": total["games"], "games_per_min": total["games"] / secs * 60, "seconds": secs, "roofline": roof,
                    "leaf_evals_per_s_per_gpu": (total["policy_evals"] + total["value_evals"]) / 2 / secs / world,
                    "per_rank_seconds": per_rank_s, "per_rank_seconds_min": min(per_rank_s), "per_rank_seconds_max": secs,
                    "games_per_rank": mine, "host_threads": threads_for(mine), "plies": total["plies"], "value_evals": total["value_evals"],
                    "value_evals_per_s": total["value_evals"] / secs,
                    "children_evaluated_per_expansion": selfplay.default_eager_top(prec, mine),
                    "black_wins": total["black_wins"], "stats_allreduce_ms": max(reduce_ms), "stats_allreduce_ms_per_rank": reduce_ms,
                    "allreduce_wait_ms_per_rank": wait_ms, "value_sums_exact": total["value_sums_exact"],
                    "first_move_hist_sum": int(sum(total["first_move_hist"])),
                    # north_star: "all-reduce visit/value statistics at the end of a generation"
                    "root_visit_hist_sum": int(sum(total["root_visit_hist"])), "root_visit_hist_top5": sorted(
                        ((go_name(m), int(n)) for m, n in enumerate(total["root_visit_hist"])), key=lambda t: -t[1])[:5],
                    "mean_root_value": total["mean_root_value"], "mean_abs_root_value": total["mean_abs_root_value"],
                    "n_root_values": total["n_root_values"], "pools": local["n_pools"], "native_loop": local["native_loop"]}

        def go_name(m):
            from bokego_amd import go
            return go.unsquash(m)

        for prec in ([args.precision] if args.no_f16x2 else [args.precision, other_name]):
            eng.set_precision(prec)
            progress(f"self-play, strong leg, {prec}: {args.selfplay_games} games over {world} rank(s)")
            sp[prec] = generation(args.selfplay_games, prec)          # STRONG scaling: the config's fixed 512 games over all ranks
        eng.set_precision(args.precision)
        # OPT-IN, labelled, never the leg the line's games_per_min is: the multi-leaf throughput mode (bk_search_params.leaves = 8: up
        # to eight rollouts of a tree wait for their values together under virtual loss -- SURVEY 7.6 allows it "only as an opt-in
        # throughput mode"; another search than the reference's, no parity claim).  What it buys grows as a rank's share shrinks.
        progress(f"self-play, strong leg in the opt-in multi-leaf mode: {args.selfplay_games} games over {world} rank(s)")
        sp["opt_in_multi_leaf"] = dict(generation(args.selfplay_games, args.precision, leaves=8), leaves=8, precision=args.precision,
                                       what="NOT the reference's search (virtual loss, 8 leaves per tree and step; off by default): the strong leg's "
                                            "games with fewer, larger requests; strength against the one-leaf search: profiles/r06_leaves_probe.txt")
        # WEAK scaling: every rank plays its own full set (what the reference's workers do: cpu_count() processes, each with
        # its own games, bin/selfplay.py:177-199) -- games x n_gpus in all, `games` per rank, the same one all-reduce at the
        # end.  At one rank it is the strong leg.  A rank's seconds here are what ONE rank needs for the whole config, so
        # both efficiencies can be read off this line: weak = min / max of per_rank_seconds against a 1-GPU run's `seconds`;
        # strong = (a rank's weak-leg seconds) / (n_gpus x the strong leg's seconds).
        if world > 1:
            progress(f"self-play, weak leg: {args.selfplay_games} games per rank")
            sp["weak"] = dict(generation(args.selfplay_games * world, args.precision), precision=args.precision,
                              what=f"{args.selfplay_games} games PER RANK ({args.selfplay_games * world} in all), one all-reduce")
            one_rank = float(np.median(sp["weak"]["per_rank_seconds"]))
            sp["strong_scaling_efficiency_vs_own_weak_leg"] = one_rank / (world * sp[args.precision]["seconds"])
            sp["weak_scaling_per_rank_seconds_min_over_max"] = sp["weak"]["per_rank_seconds_min"] / sp["weak"]["per_rank_seconds_max"]
        else:
            sp["weak"] = {"same_as": args.precision, "what": "one rank: the weak leg is the strong leg"}
        eng.set_precision(args.precision)
        st = eng.stats()      # rank 0's engine over everything above: which launch forms ran, and whether anything had to be redone
        sp["engine_counters_rank0"] = {k: st[k] for k in ("evals", "batches", "mean_batch", "positions_encoded", "split_launches", "coop_launches",
                                                           "coop_fallbacks", "f16_overflow_fallbacks", "failed_submissions", "host_wait_ms_sum")}
        sp["games_per_min"] = sp[args.precision]["games_per_min"]
        sp["stats_allreduce_ms"] = sp[args.precision]["stats_allreduce_ms"]

    # ---- the collectives are over: the other ranks leave, rank 0 finishes the line ---------------------------------------
    # north_star: "1/2/4/8-GPU self-play throughput reported next to the reference CPU path timed on the host cores (core
    # count stated) in the same run" -- so the CPU legs and the counter passes run at ANY world size, on rank 0, once the
    # other ranks have let go of the host's cores and of the GPUs (they exit here; under torch.distributed.run the agent
    # simply waits for rank 0).  Rank 0 takes the CPUs it had before it pinned itself to its slice: the CPU legs then see the
    # host exactly as a 1-GPU run does.
    ranks_seen = dist.get_world_size() if dist is not None else 1
    dist_backend = backend if dist is not None else None
    # which fabric and which cards the line was measured on: RCCL's version and every rank's device ordinal + PCI address
    rccl_version = None
    if dist is not None and backend == "nccl":
        try:
            rccl_version = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as exc:  # noqa: BLE001  (a field of the line, not a reason to lose it)
            rccl_version = f"unknown ({type(exc).__name__})"
    props = torch.cuda.get_device_properties(local_rank)
    pci_code = float((getattr(props, "pci_domain_id", 0) << 16) | (getattr(props, "pci_bus_id", 0) << 8) | getattr(props, "pci_device_id", 0))
    rank_devices = [{"rank": r, "device": int(d), "pci": f"{int(c) >> 16:04x}:{(int(c) >> 8) & 0xff:02x}:{int(c) & 0xff:02x}.0"}
                    for r, (d, c) in enumerate(zip(gather(float(local_rank)), gather(pci_code)))]
    rank0_cpus = format_cpulist(os.sched_getaffinity(0)) if "BK_BENCH_CPUS" in os.environ else None
    others = None
    if dist is not None:      # (the tensor collective the line's other fields use, not all_gather_object: nothing new for RCCL's first N > 1 run)
        others = [int(p) for p in gather(float(os.getpid())) if int(p) != os.getpid()]
    if rank != 0:
        eng.close()                              # the GPU is let go of before the group is: rank 0's counter passes come next
    if not leave_the_group(dist, rank, world, others):
        return
    dist = None

    progress("collectives over; rank 0 alone: counter passes")
    live = None
    if not args.no_live_pmc:
        live = live_counters(args.batch, args.precision, device=local_rank)

    cpu = None
    if not args.no_cpu_baseline:
        progress("CPU baselines")
        cpu = cpu_baseline(pw, vw, x_host, head_logits, head_values)
        if sp is not None:
            sp["cpu_baseline"] = selfplay_cpu_baseline(cpu["cores"])

    if rank == 0:
        value = world * args.batch * args.steps / head["dt"]
        line = {
            "metric": "leaf-evals/sec (policy+value, batched 9x9 positions)",
            "value": value, "unit": "leaf-evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["dt"] / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE[args.precision],
            "data": "synthetic (seeded random-playout positions, SURVEY 8d recipe)",
            "config": {"workload": f"configs[1]: batch={args.batch} 9x9 positions, PolicyNet logits+softmax and "
                                   "ValueNet, device-resident inputs/outputs",
                       "batch_per_gpu": args.batch, "weights": "policy_19 + value_synth (tests/golden)",
                       "precision": args.precision, "parity": head_parity,
                       "parity_workload_sample": cpu["timed_output_vs_oracle_sample"] if cpu else None,
                       "sharding": f"positions x{world}, no data-path collective"},
            "sustained": head["sustained"],
            "roofline": roofline(args.precision, args.batch, head["kernel_ms"], head["sustained"], head["p10_p50_p90"], live),
            other_name: other,
            "cpu_baseline": cpu,
            "selfplay": sp,
            "small_batch_latency": small,
            "collective_ranks_seen": ranks_seen,
            "collective_backend": dist_backend,
            "rccl_version": rccl_version,
            "rank_devices": rank_devices,
            "device_name": props.name,
            "per_rank_leaf_evals_per_s": per_rank,
            "launched_by": ("torch.distributed.run" if os.environ.get("BK_BENCH_PINNED_BY") == "rank" or
                            ("TORCHELASTIC_RUN_ID" in os.environ and "BK_BENCH_CPUS" not in os.environ) else
                            "bench.py launcher" if "BK_BENCH_CPUS" in os.environ else "direct"),
            "rank0_cpus": rank0_cpus,
            "host_buffer_e2e_leaf_evals_per_s": e2e,
            "host_buffer_e2e_u8_pipelined_leaf_evals_per_s": e2e_u8,
            "host_positions_e2e_pipelined_leaf_evals_per_s": e2e_pos,
        }
        print(json.dumps(line), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
