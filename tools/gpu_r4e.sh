#!/usr/bin/env bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
T=${1:-r4e}
timeout 600 python tools/probe_hot_streams.py 65536 > gpurun_out/${T}_hot_streams.txt 2>&1; cat gpurun_out/${T}_hot_streams.txt | tail -6 | cut -c1-300
for c in s l 4k; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --config $c > gpurun_out/${T}_bench_$c.json 2> gpurun_out/${T}_bench_$c.err; echo "bench $c rc=$?"
  python - <<PY
import json
d=json.loads(open('gpurun_out/${T}_bench_$c.json').read().strip().splitlines()[-1])
print('$c', d['ms_per_step'], d['value'], d['kernels_ms'], 'iso', d['isolated']['ms_per_step'], 'ref', d['reference_surface']['ms_per_step'], d['reference_surface']['with_compat_optimizer']['ms_per_step'])
PY
done
rm -f gpurun_out/${T}_psnr_long.jsonl
NVP_PSNR_LOG=gpurun_out/${T}_psnr_long.jsonl NVP_PARITY_REPORT=1 timeout 900 python -m pytest tests/test_gpu_parity.py::test_psnr_after_1000_steps_matches_oracle -q --timeout 800 2>&1 | tail -3
for v in "a16g05:--natural-alpha 1.6 --natural-grain 0.5" "a14g08:--natural-alpha 1.4 --natural-grain 0.8"; do
  n=${v%%:*}; f=${v#*:}
  rm -f gpurun_out/${T}_train90_$n.jsonl
  timeout 400 python -m nvp_amd.train --video natural $f --seconds 90 --eval-8bit --report-every 500 --log gpurun_out/${T}_train90_$n.jsonl > /dev/null 2> gpurun_out/${T}_train90_$n.err; echo "train $n rc=$?"; tail -1 gpurun_out/${T}_train90_$n.jsonl | cut -c1-600
done
