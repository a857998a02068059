#!/usr/bin/env bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
NVP_PARITY_REPORT=1 timeout 2000 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/r3c_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3c_pytest.log
tail -15 gpurun_out/r3c_pytest.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3c_bench.json 2> gpurun_out/r3c_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3c_bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['kernels_ms']); print('isolated', d['isolated']['ms_per_step'], d['isolated']['kernels_ms'])
PY
bash tools/gpu_prof.sh r3c > /dev/null; head -30 gpurun_out/r3c_kernel_stats.txt | cut -c1-170
bash tools/gpu_prof.sh r3c_iso NVP_EARLY_ADAMW=0 NVP_SCATTER_PRESORT=0 NVP_SAMPLER_PREFETCH=0 > /dev/null; head -30 gpurun_out/r3c_iso_kernel_stats.txt | cut -c1-170
