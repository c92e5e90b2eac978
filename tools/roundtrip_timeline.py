"""Timeline of the last 200 requests of tools/roundtrip_trace.py from rocprofv3's CSVs: medians of the gaps between the HIP
calls of a request, its kernels and its copy."""
import csv, glob, sys
from statistics import median
d = sys.argv[1]
def rows(pat):
    fs = glob.glob(f"{d}/**/*{pat}*.csv", recursive=True)
    out = []
    for f in fs:
        out += list(csv.DictReader(open(f)))
    return out
k = rows("kernel_trace")
m = rows("memory_copy_trace")
h = rows("hip_api_trace")
enc = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in k if "encode" in r["Kernel_Name"])
leaf = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in k if "leaf_eval" in r["Kernel_Name"])
cp = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in m if "DEVICE_TO_HOST" in r.get("Direction", "") .upper() or "DtoH" in r.get("Direction", ""))
print(f"{len(enc)} encoder launches, {len(leaf)} leaf launches, {len(cp)} device-to-host copies, {len(h)} HIP calls")
names = {}
for r in h:
    names.setdefault(r["Function"], []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for n, v in sorted(names.items(), key=lambda kv: -len(kv[1]))[:14]:
    print(f"  {n:34s} {len(v):6d} calls, median {median(e - s for s, e in v) / 1e3:7.2f} us")
n = min(len(enc), len(leaf), len(cp))
enc, leaf, cp = enc[-200:], leaf[-200:], cp[-200:]
sync = sorted(names.get("hipEventSynchronize", []))[-200:]
launch = sorted(names.get("hipLaunchKernel", []) + names.get("hipModuleLaunchKernel", []) + names.get("hipExtModuleLaunchKernel", []))
def med(xs): return median(xs) / 1e3
print(f"encoder kernel           {med([e - s for s, e in enc]):7.2f} us")
print(f"encoder end -> leaf start {med([l[0] - e[1] for e, l in zip(enc, leaf)]):7.2f} us")
print(f"leaf kernel              {med([e - s for s, e in leaf]):7.2f} us")
if len(cp) >= 200:
    print(f"leaf end -> copy start    {med([c[0] - l[1] for l, c in zip(leaf, cp)]):7.2f} us")
    print(f"copy                     {med([e - s for s, e in cp]):7.2f} us")
    last = cp
else:
    print("no device-to-host copy: the kernels write the outputs into the caller's pinned buffer")
    last = leaf
if len(sync) == 200:
    print(f"last device work ends -> wait returns  {med([s[1] - c[1] for c, s in zip(last, sync)]):7.2f} us")
    print(f"wait returns -> next encoder starts    {med([e[0] - s[1] for s, e in zip(sync[:-1], enc[1:])]):7.2f} us")
    first_launch = [min((l for l in launch if s0[1] <= l[0] < e0[0] + 1), default=None) for s0, e0 in zip(sync[:-1], enc[1:])]
    gaps = [e0[0] - l[0] for l, e0 in zip(first_launch, enc[1:]) if l]
    if gaps:
        print(f"  of which: first hipLaunchKernel call -> encoder starts {med(gaps):7.2f} us")
print(f"request period            {med([b[0] - a[0] for a, b in zip(enc[:-1], enc[1:])]):7.2f} us")
