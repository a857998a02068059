#!/usr/bin/env bash
# Evidence run on the GPU box: full GPU suite, bench lines (s with CPU baseline, l, 4k, eval), rocprofv3 kernel traces in step and alone.
# usage: gpurun --timeout 3600 -- 'bash tools/gpu_evidence.sh [TAG]'   -> gpurun_out/TAG_* (copy what is to be judged into profiles/)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-ev}
mkdir -p gpurun_out
NVP_PARITY_REPORT=1 timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/${T}_pytest.log; tail -4 gpurun_out/${T}_pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_s.json 2> gpurun_out/${T}_bench_s.err
python bench.py --steps 20 --warmup 5 --config l > gpurun_out/${T}_bench_l.json 2> gpurun_out/${T}_bench_l.err
python bench.py --steps 20 --warmup 5 --config 4k > gpurun_out/${T}_bench_4k.json 2> gpurun_out/${T}_bench_4k.err
python bench.py --mode eval --steps 8 --warmup 2 > gpurun_out/${T}_bench_eval.json 2> gpurun_out/${T}_bench_eval.err
for c in s l 4k; do python - <<PY
import json
d=json.loads(open('gpurun_out/${T}_bench_$c.json').read().strip().splitlines()[-1])
print('$c', d['ms_per_step'], d['value'], d['kernels_ms'], 'iso', d['isolated']['ms_per_step'], d['isolated']['kernels_ms'], 'roof', d['roofline']['kernel'], d['roofline']['frac'])
PY
done
bash tools/gpu_prof.sh ${T}_s > /dev/null
bash tools/gpu_prof.sh ${T}_s_iso NVP_EARLY_ADAMW=0 NVP_SCATTER_PRESORT=0 NVP_SAMPLER_PREFETCH=0 > /dev/null
BENCH_ARGS="--config l" bash tools/gpu_prof.sh ${T}_l > /dev/null
BENCH_ARGS="--config 4k" bash tools/gpu_prof.sh ${T}_4k > /dev/null
head -14 gpurun_out/${T}_s_iso_kernel_stats.txt | cut -c1-140
