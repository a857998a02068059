#!/usr/bin/env bash
# 5000-step, full-size (configs[1]) trajectories of the default fp16x2 build, the all-fp32-MFMA twin, and
# each started <= 1 ulp away, on identical init and batches -> gpurun_out/lh_*.jsonl + lh_compare.txt
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
STEPS=${STEPS:-5000}; EVERY=${EVERY:-250}; VIDEO=${VIDEO:-natural}
TWIN=$PWD/nvp_amd/csrc/libnvp_hip_fp32mfma.so
rm -f gpurun_out/lh_*.jsonl
run() { # tag lib ulp
  NVP_HIP_LIB=$2 timeout 900 python tools/long_horizon.py --steps $STEPS --every $EVERY --video $VIDEO --tag $1 --ulp $3 --out gpurun_out/lh_$1.jsonl > gpurun_out/lh_$1.log 2>&1 || echo "run $1 FAILED" | tee -a gpurun_out/lh_compare.txt
}
run f16x2 "" 0
run fp32mfma $TWIN 0
run fp32mfma_1ulp $TWIN 1
run f16x2_1ulp "" 1
python tools/long_horizon.py --compare gpurun_out/lh_fp32mfma.jsonl gpurun_out/lh_f16x2.jsonl gpurun_out/lh_fp32mfma_1ulp.jsonl gpurun_out/lh_f16x2_1ulp.jsonl | tee gpurun_out/lh_compare.txt
