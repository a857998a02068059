#!/usr/bin/env bash
# round-3 evidence run 2: PMC passes (s: all four; l, 4k: fetch + write), 90-s encode of the natural-statistics clip
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
bash tools/pmc.sh r03s > gpurun_out/r03s_pmc.log 2>&1
python tools/pmc_summarize.py r03s s > /dev/null 2>&1
BENCH_ARGS="--config l" PASSES="fetch write" bash tools/pmc.sh r03l > gpurun_out/r03l_pmc.log 2>&1
python tools/pmc_summarize.py r03l l > /dev/null 2>&1
BENCH_ARGS="--config 4k" PASSES="fetch write" bash tools/pmc.sh r034k > gpurun_out/r034k_pmc.log 2>&1
python tools/pmc_summarize.py r034k 4k > /dev/null 2>&1
cp profiles/pmc_traffic.json profiles/r03s_pmc_summary.txt profiles/r03l_pmc_summary.txt profiles/r034k_pmc_summary.txt gpurun_out/ 2>/dev/null
cat profiles/pmc_traffic.json
python -m nvp_amd.train --video natural --seconds 90 --eval-8bit --report-every 500 --log gpurun_out/r03_train_90s_natural.jsonl > gpurun_out/r03_train_90s_natural.log 2>&1; tail -2 gpurun_out/r03_train_90s_natural.log
