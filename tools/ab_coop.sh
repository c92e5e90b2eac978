#!/bin/bash
# A/B of the cooperative forms on one GPU box: bokego_amd/libbokego_amd_old.so (tools/build_ref_lib.sh <ref>) against the current
# build, request sizes of every form (tools/coop_probe.py, coop3_probe.py), alternating.  Output: gpurun_out/ab_coop.txt
set -e
mkdir -p gpurun_out; OUT=gpurun_out/ab_coop.txt; : > $OUT
for v in old new old new; do
  L=$PWD/bokego_amd/libbokego_amd.so; [ $v = old ] && L=$PWD/bokego_amd/libbokego_amd_old.so
  echo "=== $v: coop_probe" >> $OUT
  BK_LIB_ANY_ABI=1 BK_LIB_PATH=$L timeout -k 10 200 python tools/coop_probe.py >> $OUT 2>&1
  echo "=== $v: coop3_probe --eight" >> $OUT
  BK_LIB_ANY_ABI=1 BK_LIB_PATH=$L timeout -k 10 200 python tools/coop3_probe.py --eight >> $OUT 2>&1
done
for v in old new; do
  L=$PWD/bokego_amd/libbokego_amd.so; [ $v = old ] && L=$PWD/bokego_amd/libbokego_amd_old.so
  echo "=== $v: coop3_probe" >> $OUT
  BK_LIB_ANY_ABI=1 BK_LIB_PATH=$L timeout -k 10 200 python tools/coop3_probe.py >> $OUT 2>&1
done
