#!/usr/bin/env bash
# round-3 evidence run 1: full GPU suite, bench lines (s with CPU baseline, l, 4k, eval), kernel traces
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
NVP_PARITY_REPORT=1 timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 > gpurun_out/r3h_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3h_pytest.log; tail -4 gpurun_out/r3h_pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r3h_bench_s.json 2> gpurun_out/r3h_bench_s.err
python bench.py --steps 20 --warmup 5 --config l > gpurun_out/r3h_bench_l.json 2> gpurun_out/r3h_bench_l.err
python bench.py --steps 20 --warmup 5 --config 4k > gpurun_out/r3h_bench_4k.json 2> gpurun_out/r3h_bench_4k.err
python bench.py --mode eval --steps 8 --warmup 2 > gpurun_out/r3h_bench_eval.json 2> gpurun_out/r3h_bench_eval.err
for c in s l 4k; do python - <<PY
import json
d=json.loads(open('gpurun_out/r3h_bench_$c.json').read().strip().splitlines()[-1])
print('$c', d['ms_per_step'], d['value'], d['kernels_ms'], 'iso', d['isolated']['ms_per_step'], d['isolated']['kernels_ms'], 'roof', d['roofline']['kernel'], d['roofline']['frac'])
PY
done
bash tools/gpu_prof.sh r3h_s > /dev/null
bash tools/gpu_prof.sh r3h_s_iso NVP_EARLY_ADAMW=0 NVP_SCATTER_PRESORT=0 NVP_SAMPLER_PREFETCH=0 > /dev/null
BENCH_ARGS="--config l" bash tools/gpu_prof.sh r3h_l > /dev/null
BENCH_ARGS="--config 4k" bash tools/gpu_prof.sh r3h_4k > /dev/null
head -14 gpurun_out/r3h_s_iso_kernel_stats.txt | cut -c1-140
