#!/usr/bin/env bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
T=${1:-r4g}
NVP_PARITY_REPORT=1 timeout 1200 python -m pytest tests/test_gpu_parity.py::test_tile_fused_step_is_bit_identical_to_the_three_kernel_step tests/test_gpu_parity.py::test_psnr_tracks_the_oracle_along_a_1000_step_schedule -q --timeout 900 > gpurun_out/${T}_new.log 2>&1; echo "new rc=$?"; grep -E "^E  |passed|failed" gpurun_out/${T}_new.log | cut -c1-300 | head
grep psnr_equal_steps_windows gpurun_out/parity_report.jsonl | cut -c1-900
for rep in 1 2; do for tf in 0 1; do
NVP_TILE_FUSED=$tf python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference-surface --no-isolate 2>gpurun_out/${T}_bench_tf$tf.err | tee gpurun_out/${T}_bench_tf$tf.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tile_fused=$tf', d['ms_per_step'], d['kernels_ms'])" | tee -a gpurun_out/${T}_ab_tile_fused.txt
done; done
