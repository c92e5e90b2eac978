"""bench.py's CPU legs (the reference-CPU baselines) run without a GPU: keeps them from rotting between GPU runs."""
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
