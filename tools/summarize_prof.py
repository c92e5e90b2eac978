#!/usr/bin/env python3
"""Condense a tools/profile_bench.sh output directory into profiles/<tag>_*.{csv,md} (tracked)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(REPO, "gpurun_out", f"prof_{tag}")
dst = os.path.join(REPO, "profiles")
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(dst, f"{tag}_bench.json"), "w"), indent=1)

rows = list(csv.DictReader(open(stats)))
kern = [r for r in rows if "bk_leaf_eval" in r["Name"]][0]
pmc = collections.OrderedDict()
meta = {}
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv"))):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "bk_leaf_eval" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
    for k, v in agg.items():
        pmc[k] = (sum(v) / len(v), len(v))

avg_ms = float(kern["AverageNs"]) / 1e6
B = bench["config"]["batch_per_gpu"]
with open(os.path.join(dst, f"{tag}_summary.md"), "w") as o:
    o.write(f"# rocprofv3 summary {tag} (MI355X, `python3 bench.py`, batch {B})\n\n")
    o.write("## --kernel-trace --stats\n\n| kernel | calls | avg ms | min ms | max ms | % |\n|---|---|---|---|---|---|\n")
    for r in rows[:4]:
        o.write(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['AverageNs'])/1e6:.4f} | {float(r['MinNs'])/1e6:.4f} | "
                f"{float(r['MaxNs'])/1e6:.4f} | {r['Percentage']} |\n")
    o.write(f"\nbench.py HIP-event kernel time (un-profiled run): {bench['roofline']['kernel_ms']:.4f} ms; "
            f"rocprofv3 average: {avg_ms:.4f} ms.\n")
    o.write(f"\nDispatch: {meta}\n\n## --pmc passes (mean per dispatch of the leaf-eval kernel)\n\n| counter | mean | n |\n|---|---|---|\n")
    for k, (m, n) in pmc.items():
        o.write(f"| {k} | {m:.6g} | {n} |\n")
    g = pmc.get("GRBM_GUI_ACTIVE", (0, 0))[0]
    if g:
        o.write(f"\nEffective clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel time = {g/8/(avg_ms*1e-3)/1e9:.3f} GHz\n")
    mf = pmc.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0))[0]
    if mf and g:
        o.write(f"MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE/8) = {mf/1024/(g/8)*100:.1f} %\n")
    fs, ws = pmc.get("FETCH_SIZE", (0, 0))[0], pmc.get("WRITE_SIZE", (0, 0))[0]
    if fs:
        traffic = (2 * fs + ws) * 1024
        o.write(f"Fabric-side traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 = {traffic/1e6:.1f} MB "
                f"(gfx950 FETCH_SIZE correction x2, MI355X_MICROARCH.md HBM section); algorithmic "
                f"{bench['roofline']['algorithmic_hbm_bytes_per_launch']/1e6:.1f} MB + 7.9 MB weights. "
                f"= {traffic/(avg_ms*1e-3)/1e9:.1f} GB/s vs 8000 GB/s peak.\n")
    h, m = pmc.get("TCC_HIT_sum", (0, 0))[0], pmc.get("TCC_MISS_sum", (0, 0))[0]
    if h:
        o.write(f"L2 hit rate = {h/(h+m)*100:.2f} %\n")
    c, a = pmc.get("SQ_LDS_BANK_CONFLICT", (0, 0))[0], pmc.get("SQ_LDS_IDX_ACTIVE", (0, 0))[0]
    if a:
        o.write(f"LDS: bank-conflict cycles / active cycles = {c/a*100:.1f} %; LDS active = {a/256/(g/8)*100:.1f} % of CU time\n")
fs, ws = pmc.get("FETCH_SIZE", (0, 0))[0], pmc.get("WRITE_SIZE", (0, 0))[0]
json.dump({"tag": tag, "precision": bench["config"].get("precision"), "kernel": kern["Name"], "rocprof_avg_kernel_ms": avg_ms, "batch": B,
           "counters": {k: v[0] for k, v in pmc.items()},
           "hbm_traffic_bytes_per_launch": (2 * fs + ws) * 1024 if fs else None,
           "traffic_formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024, gfx950 FETCH_SIZE x2 correction"},
          open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)
print(open(os.path.join(dst, f"{tag}_summary.md")).read())
