#!/usr/bin/env bash
# A/B on one GPU box: workgroup-shared weight ring (NVP_MLP_RING_FWD, NVP_MLP_RING_BWD) and LDS-staged gather (NVP_ENCODE_LDS) vs the per-wave /
# global-gather kernels: (1) outputs and gradients must be BIT-identical, (2) bench stage times of each variant.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
# the kernel variants these switches select live in the experiments library (nvp_amd/csrc/build.sh)
export NVP_HIP_LIB=${NVP_HIP_LIB:-$PWD/nvp_amd/csrc/libnvp_hip_experiments.so}
mkdir -p gpurun_out
TAG=${1:-ab}
for F in 2 4; do
  for n in 100000 4099; do
    NVP_MLP_RING_FWD=0 NVP_MLP_RING_BWD=0 NVP_ENCODE_LDS=0 NVP_DW_MERGE=0 NVP_DZ_LEVEL_MAJOR=0 python tools/ab_dump.py /tmp/a_${F}_${n}.npz $F $n || exit 1
    NVP_MLP_RING_FWD=1 NVP_MLP_RING_BWD=1 NVP_ENCODE_LDS=1 NVP_DW_MERGE=1 python tools/ab_dump.py /tmp/b_${F}_${n}.npz $F $n || exit 1
    python - <<PY | tee -a gpurun_out/${TAG}_identity.txt
import numpy as np
a, b = np.load("/tmp/a_${F}_${n}.npz"), np.load("/tmp/b_${F}_${n}.npz")
bad = [k for k in a.files if not np.array_equal(a[k], b[k], equal_nan=True)]
print("F=${F} n=${n}:", "BIT-IDENTICAL (%d tensors)" % len(a.files) if not bad else "DIFFER: %s" % [(k, float(np.abs(a[k] - b[k]).max())) for k in bad])
PY
  done
done
for V in "NVP_DZ_LEVEL_MAJOR=0" "NVP_DZ_LEVEL_MAJOR=1" "NVP_DZ_LEVEL_MAJOR=0" "NVP_DZ_LEVEL_MAJOR=1"; do
  echo "== $V" | tee -a gpurun_out/${TAG}_bench.txt
  env $V python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernels_ms'])" | tee -a gpurun_out/${TAG}_bench.txt
done
