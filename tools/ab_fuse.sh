#!/bin/bash
# Fused encoding (the leaf kernel computes the planes of a small request from its position records) against the encoder kernel in front of the leaf
# kernel (BK_NO_FUSE_ENCODE=1), alternating on one box: the one-tree genmove, a rank's small shares of configs[3], the round trip of a 62-board request.
mkdir -p gpurun_out; OUT=gpurun_out/ab_fuse.txt; : > $OUT
for i in 1 2; do
  for v in fused two_kernels; do
    E=""; [ $v = two_kernels ] && E="BK_NO_FUSE_ENCODE=1"
    echo "=== $v: genmove" >> $OUT
    env $E timeout -k 10 200 python tools/genmove_probe.py 80 2>&1 | grep -E "ms/move|per call" >> $OUT
    echo "=== $v: shares" >> $OUT
    env $E timeout -k 10 300 python tools/leaves_probe.py --no-match --ab 2>&1 | grep "^world" >> $OUT
    echo "=== $v: roundtrip" >> $OUT
    env $E timeout -k 10 100 python tools/roundtrip_probe.py 2>&1 | grep -v amdgpu | tail -4 >> $OUT
  done
done
