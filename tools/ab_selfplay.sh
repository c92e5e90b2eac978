#!/bin/bash
# A/B of a self-play generation on one GPU box: old library (tools/build_ref_lib.sh) vs the current build,
# REPS alternating repetitions per game count; prints the minimum and median seconds of each.
REPS=${REPS:-5}
for g in ${GAMES:-64 128 512}; do
  for v in old new; do : > /tmp/ab_$v.txt; done
  for i in $(seq 1 $REPS); do
    for v in old new; do
      L=$PWD/bokego_amd/libbokego_amd.so; [ $v = old ] && L=$PWD/bokego_amd/libbokego_amd_old.so
      BK_LIB_PATH=$L timeout -k 10 100 python -m bokego_amd.selfplay --games $g --rollouts 400 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['seconds'])" >> /tmp/ab_$v.txt || exit 1
    done
  done
  for v in old new; do echo -n "games $g $v: "; sort -n /tmp/ab_$v.txt | python -c "import sys; v=[float(x) for x in sys.stdin]; print('min %.4f median %.4f' % (v[0], v[len(v)//2]))"; done
done
