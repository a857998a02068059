cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
PASSES="sq1 fetch write" bash tools/pmc.sh r05s > /dev/null 2>&1
PASSES="sq1 fetch write" BENCH_ARGS="--config l" bash tools/pmc.sh r05l > /dev/null 2>&1
PASSES="fetch write" BENCH_ARGS="--config 4k" bash tools/pmc.sh r054k > /dev/null 2>&1
rm -f gpurun_out/parity_report.jsonl
for s in 7 8 9; do NVP_PSNR_SEED=$s NVP_PARITY_REPORT=1 timeout 1200 python -m pytest tests/test_gpu_zz_trajectories.py -q -m gpu -k 1000_step --tb=short 2>&1 | grep "PSNR-PARITY\|passed\|failed" | cut -c1-260; done | tee gpurun_out/r5g_seeds.log
cp gpurun_out/parity_report.jsonl gpurun_out/r5g_parity_report.jsonl
ls gpurun_out/r05s_fetch gpurun_out/r05l_fetch gpurun_out/r054k_fetch | head
