#!/usr/bin/env bash
# PSNR at equal step count over a longer horizon than the unit test: HIP path vs the CPU oracle on identical batches.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; rm -f gpurun_out/psnr_track.jsonl
NVP_PSNR_STEPS=${1:-100} NVP_PSNR_LOG=gpurun_out/psnr_track.jsonl timeout 3000 python -m pytest tests/test_gpu_parity.py -q -m gpu -k psnr_at_equal 2>&1 | tail -5
tail -3 gpurun_out/psnr_track.jsonl
python - <<'PY'
import json
r=[json.loads(l) for l in open('gpurun_out/psnr_track.jsonl')]
d=[abs(x['psnr_oracle']-x['psnr_hip']) for x in r]
print("steps", len(r), "max |dPSNR|", max(d), "at step", d.index(max(d))+1, "final", r[-1])
PY
