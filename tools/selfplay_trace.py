"""One fp32 self-play generation of configs[3] -- or one rank's share of it -- for `rocprofv3 --kernel-trace --stats`:
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sp_trace_512 -- python3 tools/selfplay_trace.py 1 gpurun_out/sp_512.json
    python3 tools/selfplay_trace.py <world> <out.json>     (world = 1 / 2 / 4 / 8: this process plays rank 0's 512 / world games)
A short warm-up generation (16 games x 50 rollouts: worker threads, allocator, caches) comes first and is part of the trace
(its launches are ~1 % of the kernel time).  The JSON says what the timed generation did: seconds, steps, evaluations, launch counters;
tools/summarize_selfplay_trace.py puts the two together (kernel time / wall, time by launch form, workgroups per launch)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from bokego_amd import selfplay  # noqa: E402
from bokego_amd.bkw import load_bkw  # noqa: E402
from bokego_amd.engine import LeafEngine  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 1
out = sys.argv[2] if len(sys.argv) > 2 else None
threads = {1: 12, 2: 8, 4: 4, 8: 4}[world]
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
eng = LeafEngine(load_bkw(os.path.join(g, "policy_19.bkw")), load_bkw(os.path.join(g, "value_synth.bkw")), max_batch=8192)
ev = selfplay.EngineEvaluator(eng)
selfplay.self_play(ev, n_games=16, rollouts=50, cap=8192)
s0 = eng.stats()
local, total = selfplay.self_play(ev, n_games=512, rollouts=400, rank=0, world=world, cap=8192, threads=threads)
s1 = eng.stats()
i_val, i_pol = selfplay.STATS_FIELDS.index("value_evals"), selfplay.STATS_FIELDS.index("policy_evals")
d = {"world": world, "games": len(local["games"]), "seconds": local["seconds"], "steps": local["steps"], "pools": local["n_pools"],
     "host_threads": threads, "value_evals": float(local["local_stats"][i_val]), "policy_evals": float(local["local_stats"][i_pol]),
     "rows_requested": local["rows_requested"], "rows_sent": local["rows_sent"], "task_caps": local["task_caps"], "speculate": local["speculate"],
     "launches": {k: s1[k] - s0[k] for k in ("batches", "evals", "coop_launches", "split_launches", "positions_encoded", "coop_fallbacks")}}
flop = d["policy_evals"] * 133_413_888 + d["value_evals"] * 133_424_384
d["achieved_tflops"] = flop / d["seconds"] / 1e12
d["frac_of_fp32_mfma_peak"] = d["achieved_tflops"] / 157.3
print(json.dumps(d), flush=True)
if out:
    json.dump(d, open(out, "w"), indent=1)
eng.close()
