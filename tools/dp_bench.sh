#!/usr/bin/env bash
# Multi-GPU scaling line of bench.py on ONE node, as the driver launches it (one process per GPU over RCCL / xGMI):
#     bash tools/dp_bench.sh 8            # N = 8 (default), steps / warm-up as bench.py's defaults
#     STEPS=40 WARMUP=10 bash tools/dp_bench.sh 4
# bench.py itself refuses to print a result line if the replicas diverge: with --dp auto (default) every exchange scheme
# (reduce-scatter + sharded AdamW + all-gather, its one-hop all_to_all form, chunked all-reduce + full AdamW) runs five steps over
# RCCL and the parameters' checksums are compared across ranks (bench.py: verify_replicas), then again after the warm-up and
# after the timed steps; exit code 3 and a message on stderr instead of a JSON line if any rank differs.
set -eu
cd "$(dirname "$0")/.."
N=${1:-8}
export HSA_ENABLE_IPC_MODE_LEGACY=0
exec python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "${MASTER_PORT:-29511}" \
     bench.py --gpus "$N" --steps "${STEPS:-20}" --warmup "${WARMUP:-5}" ${BENCH_ARGS:-}
