#!/usr/bin/env bash
# A/B of environment settings on the experiments library: usage  BENCH_ARGS="--config l" ab_env.sh TAG "VAR=1 VAR2=2" "VAR=3" ...
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
TAG=$1; shift
export NVP_HIP_LIB=$PWD/nvp_amd/csrc/libnvp_hip_experiments.so
for rep in 1 2; do for V in "$@"; do
  env $V timeout 400 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-reference-surface --no-other-configs --no-isolate ${BENCH_ARGS:-} 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms']
print('%-52s step %.3f |' % ('$V', d['ms_per_step']), ' '.join('%s %.3f' % (n.replace('nvp_',''), v) for n, v in k.items()))" | tee -a gpurun_out/${TAG}_ab.txt
done; done
