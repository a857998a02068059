#!/usr/bin/env bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -x -k "mlp_golden or e2e_minus" 2>&1 | tail -2
NVP_DW_CHUNKS=304 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_real_configs.py -m gpu -q --timeout 600 -x -k "mlp_golden or real_config" 2>&1 | tail -2
for C in 256 304 384 456; do
  NVP_DW_CHUNKS=$C python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunks $C', d['ms_per_step'], d['kernels_ms']['nvp_mlp_bwd_dw'], 'iso', d['isolated']['ms_per_step'], d['isolated']['kernels_ms']['nvp_mlp_bwd_dw'])"
done
NVP_DW_CHUNKS=304 NVP_DW_CHUNKS_XF=304 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunks 304/304', d['ms_per_step'], d['kernels_ms']['nvp_mlp_bwd_dw'], 'iso', d['isolated']['ms_per_step'], d['isolated']['kernels_ms']['nvp_mlp_bwd_dw'])"
