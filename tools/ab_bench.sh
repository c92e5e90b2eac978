#!/bin/bash
# A/B on one GPU box: alternates bokego_amd/libbokego_amd_old.so (tools/build_ref_lib.sh <ref>) and the current build, 3 rounds
for i in 1 2 3; do
  for v in old new; do
    L=$PWD/bokego_amd/libbokego_amd.so; [ $v = old ] && L=$PWD/bokego_amd/libbokego_amd_old.so
    echo -n "$v: "; BK_LIB_PATH=$L timeout -k 10 300 python bench.py --no-cpu-baseline --no-selfplay --no-live-pmc --steps 300 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['roofline']['kernel_ms'])" || exit 1
  done
done
