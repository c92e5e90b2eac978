#!/bin/bash
# Sample socket power and clocks with rocm-smi while bench.py loops the hot path (evidence for the power-wall
# analysis in DESIGN.md).  usage (on the GPU box): tools/power_sample.sh <precision> <out-file>
prec=${1:-f16x2}; out=${2:-gpurun_out/power_$prec.txt}
mkdir -p "$(dirname "$out")"
python bench.py --steps ${STEPS:-8000} --warmup 10 --no-cpu-baseline --no-selfplay --precision "$prec" > "$out.bench" 2>&1 &
pid=$!
sleep 9      # torch import + engine build
: > "$out"
for i in $(seq 1 40); do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (junction|edge)" >> "$out"
  echo "--" >> "$out"
  kill -0 $pid 2>/dev/null || break
  sleep 0.25
done
wait $pid
tail -1 "$out.bench" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['config']['precision'] if 'precision' in d['config'] else '', round(d['value']), d['ms_per_step'])" >> "$out"
