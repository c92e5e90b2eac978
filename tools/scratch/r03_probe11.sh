#!/bin/bash
# process-level stress of the other entry points (fresh process each time, stop at the first that fails): the C self-play
# driver, a GTP session on the native tree, the two-engine match runner, a short bench
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/probe11
N=${1:-10}
fail() { echo "FAILED: $1 (run $2) rc=$3"; tail -30 gpurun_out/probe11/run.err; cp gpurun_out/probe11/run.err gpurun_out/probe11/failed_$1_$2.err; exit 1; }
for i in $(seq 1 $N); do
  G=$((64 + 32 * (i % 4)))
  timeout -k 10 120 ./examples/bk_selfplay tests/golden/policy_19.bkw tests/golden/value_synth.bkw $G 200 > gpurun_out/probe11/run.out 2> gpurun_out/probe11/run.err || fail c_selfplay $i $?
  grep -q moves_checksum gpurun_out/probe11/run.out || fail c_selfplay_output $i 0
  printf 'boardsize 9\nclear_board\nplay b e5\ngenmove w\ngenmove b\nanalyze 50\ngenmove w\nfinal_score\nquit\n' | timeout -k 10 120 python3 -X faulthandler -m bokego_amd.gtp -r 400 > gpurun_out/probe11/run.out 2> gpurun_out/probe11/run.err || fail gtp $i $?
  [ $(grep -c '^=' gpurun_out/probe11/run.out) -ge 8 ] || fail gtp_output $i 0
  BK_PRECISION=f16x2 timeout -k 10 120 ./examples/bk_selfplay tests/golden/policy_19.bkw tests/golden/value_synth.bkw $G 200 > gpurun_out/probe11/run.out 2> gpurun_out/probe11/run.err || fail c_selfplay_f16 $i $?
  echo "round $i ok"
done
for i in 1 2 3; do
  timeout -k 10 200 python3 -X faulthandler bench.py --steps 5 --warmup 2 --sustain 0 --no-cpu-baseline --no-live-pmc > gpurun_out/probe11/run.out 2> gpurun_out/probe11/run.err || fail bench $i $?
  grep -q '"roofline"' gpurun_out/probe11/run.out || fail bench_output $i 0
  echo "bench $i ok"
done
