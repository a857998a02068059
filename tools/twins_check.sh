#!/usr/bin/env bash
# parity subset + bench for every twin build in tools/bin (bf16 x 3 split: -DNVP_SPLIT_H2=0; fp32 MFMA: -DNVP_*_B3=0)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for lib in tools/bin/libnvp_*.so; do
  name=$(basename $lib .so)
  echo "=== $name" | tee -a gpurun_out/twins.txt
  NVP_HIP_LIB=$PWD/$lib timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "mlp_golden or forward_backward_vs_oracle or operand_split or e2e_minus or standalone_modulation" 2>&1 | tail -2 | tee -a gpurun_out/twins.txt
  NVP_HIP_LIB=$PWD/$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'], d['kernels_ms'], d['roofline']['bound'], d['roofline']['frac'])" | tee -a gpurun_out/twins.txt
done
