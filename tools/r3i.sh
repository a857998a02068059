#!/usr/bin/env bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -x -k "kernel_variants or forward_backward_vs_oracle or sorted_hint_forward" 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernels_ms']); print('iso', d['isolated']['ms_per_step'], d['isolated']['kernels_ms'])"
