#!/usr/bin/env bash
# bench stage times under a list of environment settings, interleaved twice on one box; usage: env_ab.sh TAG "A=1 B=0" "C=1" ...
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; shift
for rep in 1 2; do
  for envs in "" "$@"; do
    env $envs python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-isolate --no-reference-surface --no-other-configs --no-dp-floor --no-arithmetic-check --no-confirm 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-40s' % '${envs:-default}', d['ms_per_step'], d['kernels_ms'])" | tee -a gpurun_out/${TAG}_envab.txt
  done
done
