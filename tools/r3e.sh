#!/usr/bin/env bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export NVP_DW_GROUP=0
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_real_configs.py -m gpu -q --timeout 900 -k "not psnr and not long" > gpurun_out/r3e_pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r3e_pytest.log
for FU in 1 0; do
  NVP_FUSED_FWD=$FU python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3e_bench_f$FU.json 2> gpurun_out/r3e_bench_f$FU.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r3e_bench_f$FU.json').read().strip().splitlines()[-1])
print('FUSED=$FU', d['ms_per_step'], d['kernels_ms']); print('   isolated', d['isolated']['ms_per_step'], d['isolated']['kernels_ms'])
PY
done
python bench.py --mode eval --steps 8 --warmup 2 > gpurun_out/r3e_bench_eval.json 2> gpurun_out/r3e_bench_eval.err; python -c "
import json; d=json.loads(open('gpurun_out/r3e_bench_eval.json').read().strip().splitlines()[-1]); print({k:(v['frames_per_s'], v['stages']) for k,v in d['runs'].items()})"
