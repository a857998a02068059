#!/usr/bin/env bash
# Sample sclk / power with rocm-smi while bench.py runs (is the step power- or clock-limited?).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python bench.py --steps 2500 --warmup 10 --no-cpu-baseline > gpurun_out/clock_bench.json 2>gpurun_out/clock_bench.err &
BP=$!
for i in $(seq 1 200); do
  kill -0 $BP 2>/dev/null || break
  echo -n "$(date +%s.%N | cut -c1-14) "
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed 's/.*: //' | tr '\n' ' '
  echo
  sleep 1
done
wait $BP
cut -c1-300 gpurun_out/clock_bench.json
