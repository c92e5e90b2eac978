#!/usr/bin/env python3
"""bench.py -- leaf-evals/s of the fused PolicyNet+ValueNet HIP engine on MI355X.

A "step" is one pass of the hot path over one batch: BASELINE.json configs[1], 4096 9x9
positions, policy logits + softmax + value, inputs already resident in HBM.  With --gpus N
(launched through torch.distributed.run, one rank per GPU) every rank evaluates its own 4096
positions per step: the path shards with no data-path collective ("weak" scaling).

Prints ONE JSON line on rank 0 (contract in the task statement) carrying
  roofline      bound = MFMA: SURVEY 8d's 266,838,272 algorithmic FLOP per leaf-eval over the dense MFMA
                peak of the dtype the matrix unit executes (2,500 TFLOP/s for the default f16x2 kernel,
                157.3 for --precision f32); kernel time from HIP events on the launch stream
  cpu_baseline  the CPU oracle (oracle/nnet_ref.c, a port of the reference's forward pass) timed on this
                host's cores on a bounded sample of the same workload
  selfplay      secondary, outside the timed region: BASELINE configs[3] (512 self-play games sharded over
                the ranks + the end-of-generation all-reduce) with its own CPU baseline at N=1.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FLOP_PER_LEAF = 266_838_272          # SURVEY 8d / BASELINE.md 3: valid taps, both nets
PEAK_FP32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md chip table (dense, fp32 in / fp32 acc)
PEAK_F16_MFMA_TFLOPS = 2500.0        # MI355X_MICROARCH.md chip table (dense f16/bf16 MFMA)
BYTES_PER_LEAF = 8748 + 328          # compulsory HBM bytes (f32 planes in, 81 probs + value out)
BATCH = 4096


def make_workload(B, rank):
    """SURVEY 8d config-2 input: position i is a seeded random playout from the empty board
    (rng = default_rng(20260 + i), L = rng.integers(0, 61) uniformly random legal non-eye-filling
    moves) encoded by the build's own board engine in incremental mode -> f32 [B,27,9,9].  The recipe
    is pinned move-for-move against the reference's rules engine by tests/golden/playouts.json."""
    from bokego_amd.workload import make_batch
    return make_batch(B, seed_base=20260 + rank * B, dtype=np.float32, with_records=True)


def measured_traffic(batch):
    """HBM-side bytes per launch from the newest committed rocprofv3 PMC summary (profiles/*_pmc.json,
    produced by tools/profile_bench.sh + tools/summarize_prof.py); None if absent or another batch."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "*_pmc.json")))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    if d.get("batch") != batch:
        return None, None
    return d.get("hbm_traffic_bytes_per_launch"), os.path.basename(files[-1])


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(pw, vw, x, target_s=12.0):
    from oracle.oracle import OraclePolicy, OracleValue, set_threads

    P, V = OraclePolicy(pw), OracleValue(vw)
    # single thread first: one position at a time, and a 64-position batch
    set_threads(1)
    P(x[:1]); V(x[:1])
    t0 = time.time()
    for i in range(8):
        P(x[i:i + 1]); V(x[i:i + 1])
    b1_ms = (time.time() - t0) / 8 * 1e3
    t0 = time.time(); P(x[:64]); V(x[:64]); one_thread = 64 / (time.time() - t0)
    # a 1-GPU box gets a 16-core share of the host (more threads only fight over them)
    cores = set_threads(int(os.environ.get("BK_CPU_THREADS", min(16, len(os.sched_getaffinity(0))))))
    n = 64
    t0 = time.time(); P(x[:n]); V(x[:n]); dt = time.time() - t0  # warm + calibrate
    reps = int(max(1, min(8, target_s / max(dt * len(x) / n, 1e-3))))   # whole passes over the batch, ~target_s
    t0 = time.time()
    for _ in range(reps):
        P(x); V(x)
    dt = time.time() - t0
    return {"value": reps * len(x) / dt, "unit": "leaf-evals/s", "cores": cores, "kind": "port",
            "sample": f"{reps} passes over the same {len(x)}-position batch, oracle/nnet_ref.c (OpenMP, fp32), {dt:.1f}s",
            "cpu_model": cpu_model(), "host_cpus_visible": len(os.sched_getaffinity(0)),
            "one_thread_leaf_evals_per_s": one_thread, "one_thread_batch1_ms_per_leaf_eval": b1_ms}


def selfplay_cpu_baseline(pw, vw, plies_per_game, moves=3, rollouts=400, cores=16):
    """The reference's way of running config 4 on the host: one sequential tree per process, one position per
    network call (oracle/mcts_ref.py + the C oracle nets, 1 thread).  Bounded sample: `moves` moves of one
    game; games/min is extrapolated to `cores` independent single-thread games."""
    from oracle import oracle
    from oracle.mcts_ref import RefMCTS
    oracle.set_threads(1)
    P, V = oracle.OraclePolicy(pw), oracle.OracleValue(vw)
    t0 = time.perf_counter()
    m = RefMCTS(lambda f: P(f), lambda f: V(f))
    for _ in range(moves):
        m.rollout(rollouts)
        m.choose()
    s_per_move = (time.perf_counter() - t0) / moves
    cores = min(cores, len(os.sched_getaffinity(0)))
    oracle.set_threads(cores)
    return {"games_per_min": 60.0 / (plies_per_game * s_per_move) * cores, "unit": "games/min", "cores": cores, "kind": "port",
            "s_per_move_one_core": s_per_move,
            "sample": f"{moves} moves x {rollouts} rollouts of one sequential tree on 1 core (batch-1 network calls), "
                      f"extrapolated to {plies_per_game:.0f} plies/game and {cores} independent games"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-selfplay", action="store_true", help="skip the secondary config-4 measurement")
    ap.add_argument("--precision", choices=["f16x2", "f32"], default=os.environ.get("BK_PRECISION", "f16x2"),
                    help="conv arithmetic: fp16 hi/lo split operands with fp32 accumulation (default) or exact fp32 MFMA")
    args = ap.parse_args()

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
        sys.exit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
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

    for _ in range(args.warmup):
        eng.eval_device(x, logits=True, probs=True, value=True)
    torch.cuda.synchronize()
    eng.set_profiling(True)
    s0 = eng.stats()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = eng.eval_device(x, logits=True, probs=True, value=True)
    barrier()
    dt = time.perf_counter() - t0
    s1 = eng.stats()
    eng.set_profiling(False)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kern_ms = (s1["kernel_ms_sum"] - s0["kernel_ms_sum"]) / max(1, s1["kernel_ms_count"] - s0["kernel_ms_count"])
    assert torch.isfinite(out["value"]).all() and torch.isfinite(out["probs"]).all()

    # per-launch spread (median and p10/p90), one launch at a time, outside the timed region
    eng.set_profiling(True)
    each = []
    for _ in range(min(100, max(10, args.steps))):
        eng.eval_device(x, logits=True, probs=True, value=True)
        torch.cuda.synchronize()
        each.append(eng.stats()["last_kernel_ms"])
    eng.set_profiling(False)
    p10, p50, p90 = (float(np.percentile(each, q)) for q in (10, 50, 90))

    # PCIe-inclusive rates through the host-buffer ABI (reported beside, never as `value`):
    # (a) synchronous bk_eval with f32 planes, as the reference's host tensors would arrive;
    # (b) what the ABI is built for: uint8 planes, three tickets in flight (the engine runs H2D, kernels
    #     and D2H on three streams chained by events).
    e2e = e2e_u8 = e2e_pos = None
    if rank == 0:
        eng.eval(x_host, logits=False, probs=True, value=True)
        t1 = time.perf_counter()
        for _ in range(3):
            eng.eval(x_host, logits=False, probs=True, value=True)
        e2e = 3 * args.batch / (time.perf_counter() - t1)
        x_u8 = x_host.astype(np.uint8)
        eng.wait(eng.submit(x_u8, logits=False, probs=True, value=True))
        t1 = time.perf_counter()
        pend, n_e2e = [], 32
        for _ in range(n_e2e):                       # three tickets in flight: H2D / kernel / D2H streams overlap
            pend.append(eng.submit(x_u8, logits=False, probs=True, value=True))
            if len(pend) == 3:
                eng.wait(pend.pop(0))
        while pend:
            eng.wait(pend.pop(0))
        e2e_u8 = n_e2e * args.batch / (time.perf_counter() - t1)
        # (c) 192-byte position records in, planes encoded on the GPU (bk_submit_positions)
        ref = eng.eval(x_u8, logits=False, probs=True, value=True)
        got = eng.wait(eng.submit_positions(x_recs, logits=False, probs=True, value=True))
        assert np.array_equal(ref["probs"], got["probs"]) and np.array_equal(ref["value"], got["value"])
        t1 = time.perf_counter()
        for _ in range(n_e2e):
            pend.append(eng.submit_positions(x_recs, logits=False, probs=True, value=True))
            if len(pend) == 3:
                eng.wait(pend.pop(0))
        while pend:
            eng.wait(pend.pop(0))
        e2e_pos = n_e2e * args.batch / (time.perf_counter() - t1)

    # Secondary measurement (outside the timed region above): BASELINE config 4 -- 512 self-play games,
    # 400 rollouts/move, sharded over the ranks (gid % world), one all-reduce of the statistics at the end.
    sp = None
    if not args.no_selfplay:
        from bokego_amd import selfplay
        ev = selfplay.EngineEvaluator(eng)
        barrier()
        threads = max(1, min(16, len(os.sched_getaffinity(0)) // world))  # host cores are shared by the ranks
        local, total = selfplay.self_play(ev, n_games=512, rollouts=400, rank=rank, world=world, cap=8192, threads=threads,
                                          reduce_device=torch.device("cuda", local_rank) if backend == "nccl" else None)
        secs = local["seconds"]
        if dist is not None:
            t = torch.tensor([secs], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            secs = float(t.item())
        sp = {"config": "configs[3]: 512 games, 400 rollouts/move, games sharded gid % n_gpus", "games": total["games"],
              "games_per_min": total["games"] / secs * 60, "seconds": secs, "plies": total["plies"],
              "value_evals_per_s": total["value_evals"] / secs, "black_wins": total["black_wins"],
              "stats_allreduce_ms": local["allreduce_s"] * 1e3, "host_threads_per_rank": threads, "collective": "1 all-reduce of 89 doubles per generation"}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(pw, vw, x_host)
        if sp is not None:
            sp["cpu_baseline"] = selfplay_cpu_baseline(pw, vw, sp["plies"] / max(1.0, sp["games"]))

    if rank == 0:
        value = world * args.batch * args.steps / dt
        achieved = args.batch * FLOP_PER_LEAF / (kern_ms * 1e-3) / 1e12
        traffic, traffic_src = measured_traffic(args.batch)
        f16 = args.precision == "f16x2"
        # f16x2 executes 3 half-precision MFMAs per algorithmic (fp32-equivalent) MAC; the roofline is
        # priced strictly: algorithmic FLOP over the dense MFMA peak of the dtype the matrix unit runs.
        peak = PEAK_F16_MFMA_TFLOPS if f16 else PEAK_FP32_MFMA_TFLOPS
        assert eng.stats()["f16_device_overflow"] == 0
        line = {
            "metric": "leaf-evals/sec (policy+value, batched 9x9 positions)",
            "value": value, "unit": "leaf-evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16x2 (fp16 hi/lo split operands = 22-bit significands, fp32 accumulate)" if f16 else "f32",
            "data": "synthetic (seeded random-playout positions, SURVEY 8d recipe)",
            "config": {"workload": f"configs[1]: batch={args.batch} 9x9 positions, PolicyNet logits+softmax and "
                                   "ValueNet, device-resident inputs/outputs",
                       "batch_per_gpu": args.batch, "weights": "policy_19 + value_synth (tests/golden)",
                       "precision": args.precision, "parity": "max |dlogit| 5.3e-5 vs reference goldens (tol 1e-4)",
                       "sharding": f"positions x{world}, no data-path collective"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "bk_leaf_eval_f16_kernel<3>" if f16 else "bk_leaf_eval_kernel<3>",
                         "kernel_ms": kern_ms,
                         "kernel_ms_isolated_p10_p50_p90": [p10, p50, p90],
                         "algorithmic_flop_per_launch": args.batch * FLOP_PER_LEAF,
                         "algorithmic_hbm_bytes_per_launch": args.batch * BYTES_PER_LEAF,
                         "executed_mfma_flop_per_algorithmic_flop": 3.0 if f16 else 1.0,
                         # what the matrix unit really issued: per 3-board task and net 41,024 MFMAs of
                         # 32x32x16 (3x3 layers: 3 products, zero-halo taps skipped; layer 0: 2 products)
                         "executed_mfma_tflops": (2 * ((args.batch + 2) // 3) * 41024 * 32768 / (kern_ms * 1e-3) / 1e12)
                         if f16 and args.batch >= 768 else None,
                         "frac_of_fp32_mfma_peak": achieved / PEAK_FP32_MFMA_TFLOPS},
            "cpu_baseline": cpu,
            "selfplay": sp,
            "host_buffer_e2e_leaf_evals_per_s": e2e,
            "host_buffer_e2e_u8_pipelined_leaf_evals_per_s": e2e_u8,
            "host_positions_e2e_pipelined_leaf_evals_per_s": e2e_pos,
        }
        print(json.dumps(line), flush=True)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
