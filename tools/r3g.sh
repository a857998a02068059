#!/usr/bin/env bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for V in glabl1 glabl2; do
  NVP_HIP_LIB=$PWD/tools/bin/libnvp_$V.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-isolate 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V', d['ms_per_step'], d['kernels_ms'])"
done
