#!/usr/bin/env bash
# bench stage times for the default library and every tools/bin/libnvp_*.so variant, interleaved twice on one box
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-libs}
for rep in 1 2; do
  for lib in default tools/bin/libnvp_*.so; do
    if [ "$lib" = default ]; then unset NVP_HIP_LIB; else export NVP_HIP_LIB=$PWD/$lib; fi
    python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-28s' % '$lib', d['ms_per_step'], d['kernels_ms'])" | tee -a gpurun_out/${TAG}_bench.txt
  done
done
