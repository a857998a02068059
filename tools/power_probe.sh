#!/usr/bin/env bash
# Board power / sclk while ONE kernel stage loops (is that stage power-limited?).  usage: power_probe.sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for stage in fwd bwd dw; do
  NVP_LOOP_STAGE=$stage NVP_LOOP_SECONDS=9 python tools/ablate_mlp.py > gpurun_out/power_$stage.log 2>&1 &
  BP=$!
  sleep 5     # imports + setup
  echo "== $stage"
  for i in 1 2 3 4 5 6; do
    kill -0 $BP 2>/dev/null || break
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed 's/.*: //' | tr '\n' ' '; echo
    sleep 1
  done
  wait $BP
  tail -1 gpurun_out/power_$stage.log
done
