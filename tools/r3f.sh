#!/usr/bin/env bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_real_configs.py -m gpu -q --timeout 600 -x -k "mlp_golden or forward_backward_vs_oracle or operand_split or e2e_minus or real_config or kink or full_batch or linear" > gpurun_out/r3f_pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r3f_pytest.log
for G in 1 0; do
  NVP_DW_GLDS=$G python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3f_bench_g$G.json 2> gpurun_out/r3f_bench_g$G.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r3f_bench_g$G.json').read().strip().splitlines()[-1])
print('GLDS=$G', d['ms_per_step'], d['kernels_ms'], 'loss', d['final_loss']); print('   isolated', d['isolated']['ms_per_step'], d['isolated']['kernels_ms'])
PY
done
bash tools/gpu_prof.sh r3f_iso NVP_EARLY_ADAMW=0 NVP_SCATTER_PRESORT=0 NVP_SAMPLER_PREFETCH=0 > /dev/null; head -16 gpurun_out/r3f_iso_kernel_stats.txt | cut -c1-150
