"""Kernel trace of a self-play generation (tools/selfplay_trace.py under rocprofv3 --kernel-trace --stats) + the generation's own JSON
-> a csv of the per-kernel rows and a ten-line summary: kernel time / wall, time by launch form, workgroups per launch against the
chip's 256 CUs (how full the rounds are), achieved FLOP/s of the leg.
    python3 tools/summarize_selfplay_trace.py <trace dir> <generation.json> <profiles/out.csv> >> profiles/summary.md"""
import collections
import csv
import glob
import json
import math
import os
import re
import sys

trace_dir, gen_json, out_csv = sys.argv[1:4]
gen = json.load(open(gen_json))
tf = glob.glob(os.path.join(trace_dir, "*", "*_kernel_trace.csv"))[0]
rows = list(csv.DictReader(open(tf)))
by = collections.defaultdict(list)
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"^void |\(bk_eval_args\)$|\(.*\)$", "", name)
    wg = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    by[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), wg))
tot = sum(ns for v in by.values() for ns, _ in v)
t_first = min(int(r["Start_Timestamp"]) for r in rows)
t_last = max(int(r["End_Timestamp"]) for r in rows)
with open(out_csv, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MeanWorkgroups", "MeanRoundsOf256CUs", "MeanFillOfItsRounds"])
    for name, v in sorted(by.items(), key=lambda kv: -sum(ns for ns, _ in kv[1])):
        ns = sum(a for a, _ in v)
        wgs = [b for _, b in v]
        rounds = [max(1, math.ceil(b / 256)) for b in wgs]
        w.writerow([name, len(v), ns, round(ns / len(v)), round(100 * ns / tot, 2), round(sum(wgs) / len(v), 1), round(sum(rounds) / len(v), 2),
                    round(sum(b / (256 * r) for b, r in zip(wgs, rounds)) / len(v), 3)])
leaf = {k: v for k, v in by.items() if "bk_leaf_eval" in k}
leaf_ns = sum(ns for v in leaf.values() for ns, _ in v)
print(f"### {os.path.basename(gen_json)}: {gen['games']} games of rank 0 of {gen['world']} ({gen['pools']} pools, {gen['host_threads']} host threads)")
print(f"* generation {gen['seconds']:.3f} s, {gen['steps']} steps of {gen['rows_sent'] / max(1, gen['steps']):.0f} rows; value evaluations {gen['value_evals']:.0f}, policy {gen['policy_evals']:.0f} "
      f"-> {gen['achieved_tflops']:.1f} TFLOP/s algorithmic = **{gen['frac_of_fp32_mfma_peak']:.3f} of the fp32-MFMA peak** end to end (under the profiler)")
print(f"* kernel time in the trace (warm-up generation included): {tot / 1e9:.3f} s, of which leaf kernels {leaf_ns / 1e9:.3f} s; span of the trace {(t_last - t_first) / 1e9:.3f} s; "
      f"leaf-kernel time / generation seconds = {leaf_ns / 1e9 / gen['seconds']:.2f}")
for name, v in sorted(leaf.items(), key=lambda kv: -sum(ns for ns, _ in kv[1])):
    ns = sum(a for a, _ in v)
    wgs = [b for _, b in v]
    rounds = [max(1, math.ceil(b / 256)) for b in wgs]
    print(f"  * `{name}`: {len(v)} launches, {100 * ns / leaf_ns:.1f} % of the leaf-kernel time, {ns / len(v) / 1e3:.0f} us each, {sum(wgs) / len(v):.0f} workgroups per launch "
          f"= {sum(b / (256 * r) for b, r in zip(wgs, rounds)) / len(v):.2f} of its {sum(rounds) / len(v):.2f} round(s) of 256 CUs")
