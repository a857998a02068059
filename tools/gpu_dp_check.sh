#!/usr/bin/env bash
# data-parallel code paths on ONE GPU: the 2-rank shared-GPU tests, then forced-collective single-rank bench runs of every exchange
# scheme with the early update / sparse-first start on and off (the final losses must agree bit for bit)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
TAG=${1:-dp}
timeout 900 python -m pytest tests/test_gpu_dp2.py -m gpu -q --timeout 400 2>&1 | tail -3
for M in sharded a2a replicated; do
  for E in 1 0; do
    NVP_DP_EARLY_UPDATE=$E NVP_DP_FORCE_COLLECTIVES=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --dp $M 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$M early_update=$E', d['ms_per_step'], 'loss', repr(d['final_loss']), 'post_bwd', d['dp']['post_backward_ms_per_rank'])" | tee -a gpurun_out/${TAG}_dp_check.txt
  done
done
