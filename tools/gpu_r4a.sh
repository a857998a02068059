#!/usr/bin/env bash
# round-4 first check: the new tests, then the whole GPU suite, then the driver's bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
T=${1:-r4a}
NVP_PARITY_REPORT=1 timeout 1500 python -m pytest tests/test_gpu_twin.py tests/test_gpu_dp2.py::test_bench_plain_launch_starts_its_own_ranks tests/test_gpu_dp2.py::test_bench_plain_launch_refuses_more_ranks_than_devices "tests/test_gpu_parity.py::test_row_order_is_the_stable_sort_by_the_row_key" tests/test_gpu_parity.py::test_auto_sort_returns_rows_in_caller_order tests/test_gpu_parity.py::test_psnr_after_1000_steps_matches_oracle -q --timeout 1200 --durations=10 > gpurun_out/${T}_new.log 2>&1; echo "new rc=$?"; tail -25 gpurun_out/${T}_new.log | cut -c1-250
python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_s.json 2> gpurun_out/${T}_bench_s.err; echo "bench rc=$?"; tail -3 gpurun_out/${T}_bench_s.err
python - <<PY
import json
d=json.loads(open('gpurun_out/${T}_bench_s.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['kernels_ms'], 'iso', d['isolated']['ms_per_step'], d['isolated']['kernels_ms'])
print('ref', d['reference_surface'])
print('cpu', d['cpu_baseline'])
PY
NVP_PARITY_REPORT=1 timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 --deselect tests/test_gpu_parity.py::test_psnr_after_1000_steps_matches_oracle --deselect tests/test_gpu_twin.py > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/${T}_pytest.log; tail -5 gpurun_out/${T}_pytest.log | cut -c1-250
