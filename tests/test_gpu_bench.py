"""-m gpu: bench.py's one-line contract on a short run (the driver's invocation with fewer steps and without the CPU and
self-play legs): the keys the driver reads, the headline's arithmetic, roofline consistency, parity measured in the run."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def test_bench_line_contract():
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--sustain", "0.3",
                          "--no-cpu-baseline", "--selfplay-games", "64"], capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "bench.py prints exactly one line"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "leaf-evals/s" and d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1
    assert d["dtype"] == "f32" and d["scaling"] == "weak" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    B = d["config"]["batch_per_gpu"]
    assert abs(d["value"] - B * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.5 < r["frac"] < 1.0
    assert abs(r["achieved"] - r["algorithmic_flop_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e12) < 1e-6 * r["achieved"]
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.01
    # what the matrix unit executes per algorithmic FLOP: counted from the kernel's tile tables, not asserted
    assert 1.05 < r["executed_mfma_flop_per_algorithmic_flop"] < 1.10 and "tile tables" in r["executed_ratio_source"]
    assert abs(r["executed_frac_of_peak"] - r["frac"] * r["executed_mfma_flop_per_algorithmic_flop"]) < 1e-9
    if r.get("pmc_executed_mfma_flop_per_workgroup"):      # the committed PMC pass counted the same work per workgroup
        assert abs(r["pmc_executed_mfma_flop_per_workgroup"] / r["tile_table_mfma_flop_per_3_board_workgroup"] - 1) < 0.01
    assert r["mfma_busy"] is None or 0.8 < r["mfma_busy"] <= 1.0
    # the counters are measured in the run itself (rocprofv3 --pmc passes over a child issuing the same launch)
    if r["pmc_file_fallback"] is None:
        assert r["pmc_source"].startswith("measured in this run") and r["traffic_source"] == r["pmc_source"]
        assert 40e6 < r["traffic"] < 300e6 and 0.9 < r["mfma_busy"] < 1.0 and 2.1 < r["effective_clock_ghz"] < 2.45
        assert abs(r["pmc_executed_mfma_flop_per_launch"] / r["executed_mfma_flop_per_launch"] - 1) < 0.01     # counters = tile tables
    else:       # no rocprofv3 on this box (or a pass failed): the line says where its counters come from instead
        assert r["pmc_source"].startswith("from_file:") and r["pmc_file_fallback"] == r["pmc_source"]
    h = r["hbm"]                                            # north_star's wording: HBM GB/s against the chip's peak
    assert h["peak_GBps"] == 8000.0 and abs(h["algorithmic_GBps"] - r["algorithmic_hbm_bytes_per_launch"] / r["kernel_ms"] / 1e6) < 1e-6
    assert h["measured_GBps"] is None or (h["algorithmic_GBps"] < h["measured_GBps"] < 100 and h["frac_measured"] < 0.02)
    par = d["config"]["parity"]
    assert par["max_abs_dlogit"] < 1e-4 and par["max_abs_dprob"] < 1e-5 and par["max_abs_dvalue"] < 1e-4
    f16 = d["f16x2"]
    assert f16["roofline"]["peak"] == 2500.0 and f16["value"] > d["value"]
    sb = d["small_batch_latency"]
    assert sb["B62"]["cooperative_us"] < 0.6 * sb["B62"]["one_cu_per_board_us"] and sb["B62"]["fallbacks"] == 0
    # the self-play leg on the roofline, recomputable from the line alone
    assert d["device_name"] and d["rank_devices"][0]["rank"] == 0 and len(d["rank_devices"][0]["pci"]) >= 12
    for leg in ("f32", "f16x2"):
        g = d["selfplay"][leg]
        ro = g["roofline"]
        flop = ro["policy_evals"] * 133_413_888 + ro["value_evals"] * 133_424_384
        assert ro["policy_evals"] > 0 and ro["value_evals"] == g["value_evals"] and abs(ro["algorithmic_flop"] - flop) < 1e-6 * flop
        assert abs(ro["achieved_tflops"] - flop / g["seconds"] / 1e12) < 1e-6 * ro["achieved_tflops"]
        assert ro["peak"] == (157.3 if leg == "f32" else 2500.0) and abs(ro["frac"] - ro["achieved_tflops"] / ro["peak"]) < 1e-9
        assert 0.01 < ro["frac"] < 1.0 and len(ro["per_rank_frac"]) == 1 and 0.5 < ro["rows_sent_over_rows_requested"] <= 1.0
        assert g["stats_allreduce_ms"] == 0 and g["allreduce_wait_ms_per_rank"] == [0.0] and g["value_sums_exact"] is True
    assert d["selfplay"]["weak"]["same_as"] == "f32"
    ml = d["selfplay"]["opt_in_multi_leaf"]      # labelled, beside the line's own leg: never what games_per_min reports
    assert ml["leaves"] == 8 and "NOT the reference's search" in ml["what"] and ml["games"] == 64 and ml["roofline"]["frac"] > 0
    assert d["selfplay"]["games_per_min"] == d["selfplay"]["f32"]["games_per_min"]


def test_bench_starts_its_own_ranks():
    """`python3 bench.py --gpus 2` with no torch.distributed.run around it: the parent starts two
    fresh ranks and relays one line.  On a one-GPU box the two ranks share the card and meet over gloo (RCCL refuses two
    ranks on one device), which exercises everything except RCCL itself; with two GPUs it is the real thing."""
    import torch
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    real = torch.cuda.device_count() >= 2
    if not real:
        env.update(BK_BENCH_BACKEND="gloo", BK_BENCH_DEVICE="0")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--sustain", "0",
                          "--no-f16x2", "--selfplay-games", "32"], capture_output=True, text=True, timeout=900, cwd=REPO, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["collective_ranks_seen"] == 2 and d["collective_backend"] == ("nccl" if real else "gloo")
    assert len(d["per_rank_leaf_evals_per_s"]) == 2 and all(v > 1e4 for v in d["per_rank_leaf_evals_per_s"])
    assert d["launched_by"] == "bench.py launcher"
    # the N-rank line is complete: once the collectives are over the other ranks leave and rank 0 times
    # the reference's CPU path on the host's cores and takes the counters, as a 1-GPU run does
    cpu = d["cpu_baseline"]
    assert cpu and cpu["value"] > 100 and cpu["cores"] >= 1 and cpu["kind"] == "port"
    assert d["config"]["parity_workload_sample"]["max_abs_dlogit_vs_oracle"] < 2e-4
    assert d["selfplay"]["cpu_baseline"]["games_per_min"] > 0 and d["selfplay"]["cpu_baseline"]["cores"] == cpu["cores"]
    r = d["roofline"]
    if r["pmc_file_fallback"] is None:
        assert r["traffic_source"].startswith("measured in this run") and 40e6 < r["traffic"] < 300e6
    sp = d["selfplay"]
    assert sp["f32"]["games"] == 32 and sp["games_per_min"] > 0 and sp["stats_allreduce_ms"] > 0
    assert sp["f32"]["first_move_hist_sum"] == 32
    # visit / value statistics in the reduced vector, what each rank needed, and a WEAK leg beside the strong one
    f = sp["f32"]
    assert f["n_root_values"] == f["plies"] and f["root_visit_hist_sum"] > 380 * f["plies"] and 0 < f["mean_abs_root_value"] < 1
    assert len(f["per_rank_seconds"]) == 2 and f["per_rank_seconds_min"] <= f["per_rank_seconds_max"] == f["seconds"]
    assert f["games_per_rank"] == 16 and len(f["stats_allreduce_ms_per_rank"]) == 2 and f["native_loop"]
    # the collective timed apart from the wait for the slowest rank; which fabric, which cards
    assert len(f["allreduce_wait_ms_per_rank"]) == 2 and all(v >= 0 for v in f["allreduce_wait_ms_per_rank"]) and f["stats_allreduce_ms"] < 200
    assert len(f["roofline"]["per_rank_frac"]) == 2 and f["roofline"]["peak"] == 2 * 157.3
    assert [x["rank"] for x in d["rank_devices"]] == [0, 1] and (d["rccl_version"] is not None) == real
    w = sp["weak"]
    assert w["games"] == 64 and w["games_per_rank"] == 32 and w["first_move_hist_sum"] == 64 and len(w["per_rank_seconds"]) == 2
    assert 0 < sp["strong_scaling_efficiency_vs_own_weak_leg"] < 1.5 and 0 < sp["weak_scaling_per_rank_seconds_min_over_max"] <= 1
    if r["pmc_file_fallback"] is None:       # the counter child's own kernel time: the factors multiply to the fraction AT that time
        assert abs(r["frac_from_pmc_factors"] - r["frac_at_pmc_kernel_ms"]) < 0.02 * r["frac_at_pmc_kernel_ms"]
    B = d["config"]["batch_per_gpu"]
    assert abs(d["value"] - 2 * B * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
