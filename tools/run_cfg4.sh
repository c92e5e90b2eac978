#!/bin/bash
# BASELINE configs[4] as written: "boke.py GTP vs GnuGo, -r 1600, HIP backend, 100-game match -- win-rate and ms/move vs CPU
# baseline" (reference: bokego/gtp.py:450-604 GTP_match, boke.py:15-45; its ten recorded games: data/bokevgnugo/*.sgf).
# Needs a `gnugo` binary on PATH (neither the build container nor the pool's GPU boxes have one): exits 3 without it.
#   usage: tools/run_cfg4.sh [tag=r03] [games=100] [rollouts=1600] [cpu_games=20]
# Writes profiles/<tag>_match_vs_gnugo.json (HIP backend, in-process) and profiles/<tag>_match_vs_gnugo_cpu_backend.json
# (the same search on the reference's torch-CPU operators, oracle/gtp_cpu.py, as the "CPU baseline" for ms/move), SGFs under
# gpurun_out/cfg4_<tag>/.  The value head shipped here is synthetic (no trained ValueNet ships with the reference), so the
# win-rate says how the *bare trained policy + a noise value head* fares, not how bokego v0.3 did (10-0, report.pdf III-C).
set -o pipefail
TAG=${1:-r03}; GAMES=${2:-100}; R=${3:-1600}; CPU_GAMES=${4:-20}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
GNUGO=${GNUGO:-$(command -v gnugo)}
if [ -z "$GNUGO" ]; then echo "run_cfg4: no gnugo binary on PATH (set GNUGO=/path/to/gnugo)"; exit 3; fi
OPP="$GNUGO --mode gtp --chinese-rules --level 10 --boardsize 9 --komi 5.5"
cd "$ROOT" && mkdir -p gpurun_out/cfg4_$TAG profiles
python3 -m bokego_amd.match --games "$GAMES" -r "$R" --opponent "$OPP" --opponent-name gnugo \
    --sgf gpurun_out/cfg4_$TAG/hip --json-out profiles/${TAG}_match_vs_gnugo.json || exit 1
python3 -m bokego_amd.match --games "$CPU_GAMES" --engine "python3 -m oracle.gtp_cpu -r $R" --engine-name boke-cpu-r$R \
    --opponent "$OPP" --opponent-name gnugo --sgf gpurun_out/cfg4_$TAG/cpu --json-out profiles/${TAG}_match_vs_gnugo_cpu_backend.json || exit 1
python3 - <<PY
import json
h = json.load(open("profiles/${TAG}_match_vs_gnugo.json")); c = json.load(open("profiles/${TAG}_match_vs_gnugo_cpu_backend.json"))
print(json.dumps({"config": "configs[4]", "hip": {"games": h["games"], "win_rate": h["win_rate"], "ms_per_move": h["ms_per_move"]},
                  "cpu_backend": {"games": c["games"], "win_rate": c["win_rate"], "ms_per_move": c["ms_per_move"]}}))
PY
