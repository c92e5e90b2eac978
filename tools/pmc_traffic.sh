#!/bin/bash
# Where the fabric-side bytes of a B = 4,096 step come from, and how much they move from run to run (VERDICT r4 next #6:
# 178 MB in profiles/r04_bench_final.json, 188 MB in the driver's run of the same commit).  Separate rocprofv3 --pmc passes
# (MI355X_MICROARCH.md, HBM section) over the child bench.py itself uses (`--pmc-child`: the timed launch, 8 times), each pass
# REPEATS times; per dispatch, per kernel (the ten 3-board rounds <3,false> and the 2-board tail <2,false>).
#   tools/pmc_traffic.sh [repeats=3]      (GPU box; output: gpurun_out/pmc_traffic/summary.txt)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/pmc_traffic; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd $ROOT
REPEATS=${1:-3}
for R in $(seq 1 $REPEATS); do
  for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-40)_r$R
    rocprofv3 --pmc $C --output-format csv -d $OUT/$N -- python3 bench.py --pmc-child --batch 4096 --precision f32 > $OUT/$N.log 2>&1 || echo "pass $N failed" >> $OUT/summary.txt
  done
done
python3 - <<PY | tee -a $OUT/summary.txt
import csv, glob, collections, re
per = collections.defaultdict(list)          # (counter, kernel kind, repeat) -> per-dispatch values in dispatch order
for f in sorted(glob.glob('$OUT/*/*/*_counter_collection.csv')):
    rep = int(re.search(r'_r(\d+)/', f).group(1))
    rows = [r for r in csv.DictReader(open(f)) if 'bk_leaf_eval_kernel' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    for r in rows:
        kind = '3-board rounds' if '<3' in r['Kernel_Name'] else '2-board tail' if '<2' in r['Kernel_Name'] else r['Kernel_Name'][:40]
        per[(r['Counter_Name'], kind, rep)].append(float(r['Counter_Value']))
print('# per dispatch (8 launches of the step per pass), by counter / kernel / repeat')
for (c, kind, rep), v in sorted(per.items()):
    print(f'{c:24s} {kind:15s} run {rep}: ' + ' '.join(f'{x:,.0f}' for x in v))
print('# fabric-side MB per step = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 / 1e6 (gfx950 correction), mean over dispatches 2..8, per repeat')
reps = sorted({k[2] for k in per})
for rep in reps:
    tot = 0.0
    parts = []
    for kind in ('3-board rounds', '2-board tail'):
        f = per.get(('FETCH_SIZE', kind, rep), [])[1:]
        w = per.get(('WRITE_SIZE', kind, rep), [])[1:]
        if f and w:
            mb = (2 * sum(f) / len(f) + sum(w) / len(w)) * 1024 / 1e6
            first = (2 * per[('FETCH_SIZE', kind, rep)][0] + per[('WRITE_SIZE', kind, rep)][0]) * 1024 / 1e6
            parts.append(f'{kind} {mb:.1f} MB (first dispatch {first:.1f})')
            tot += mb
    print(f'run {rep}: {tot:.1f} MB per step = ' + ' + '.join(parts))
for kind in ('3-board rounds', '2-board tail'):
    for rep in reps:
        h, m = per.get(('TCC_HIT_sum', kind, rep), []), per.get(('TCC_MISS_sum', kind, rep), [])
        rd, rd32 = per.get(('TCC_EA_RDREQ_sum', kind, rep), []), per.get(('TCC_EA_RDREQ_32B_sum', kind, rep), [])
        if h and m and rd:
            print(f'{kind:15s} run {rep}: L2 hit rate {sum(h) / (sum(h) + sum(m)):.4f}; misses per dispatch {sum(m) / len(m):,.0f}; '
                  f'EA read requests per dispatch {sum(rd) / len(rd):,.0f} of which 32-byte {sum(rd32) / max(1, len(rd32)):,.0f} '
                  f'-> {(64 * (sum(rd) / len(rd) - sum(rd32) / max(1, len(rd32))) + 32 * sum(rd32) / max(1, len(rd32))) / 1e6:.1f} MB read')
PY
