#!/usr/bin/env bash
# short, bounded diagnosis stages; every stage has its own timeout and log under gpurun_out/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
# the kernel variants these switches select live in the experiments library (nvp_amd/csrc/build.sh)
export NVP_HIP_LIB=${NVP_HIP_LIB:-$PWD/nvp_amd/csrc/libnvp_hip_experiments.so}
mkdir -p gpurun_out
TAG=${1:-diag}
export NVP_PARITY_REPORT=1
stage() { # name timeout cmd...
  local name=$1 to=$2; shift 2
  echo "=== $name" | tee -a gpurun_out/${TAG}_summary.txt
  ( timeout $to "$@" ) > gpurun_out/${TAG}_$name.log 2>&1
  echo "$name rc=$?" | tee -a gpurun_out/${TAG}_summary.txt
  tail -5 gpurun_out/${TAG}_$name.log | cut -c1-300 | tee -a gpurun_out/${TAG}_summary.txt
}
stage mlp_old 400 env NVP_MLP_RING_FWD=0 NVP_MLP_RING_BWD=0 NVP_ENCODE_LDS=0 python -m pytest tests/test_gpu_parity.py -x -v --tb=short --timeout 150 -k "mlp_golden or standalone or e2e_minus"
stage ab 900 bash tools/ab_ring.sh ${TAG}
stage mlp_new 400 python -m pytest tests/test_gpu_parity.py -x -v --tb=short --timeout 150 -k "mlp_golden or standalone or e2e_minus or lds_staged or nvp_forward_backward"
stage dp2 500 python -m pytest tests/test_gpu_dp2.py -x -v --tb=short --timeout 240 -k "whole_batch"
