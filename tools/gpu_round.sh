#!/usr/bin/env bash
# One GPU-box round: parity tests, smoke, bench, rocprof kernel stats. Logs -> gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r}
{ rocm-smi --showproductname 2>/dev/null | head -8; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket"; } > gpurun_out/${TAG}_box.txt 2>&1
timeout 2700 python -m pytest tests -m gpu -q --tb=short --timeout 1200 --durations=15 > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/${TAG}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/${TAG}_smoke.log
timeout 900 python bench.py --steps ${STEPS:-10} --warmup 3 > gpurun_out/${TAG}_bench.log 2>&1; echo "bench rc=$?" | tee -a gpurun_out/${TAG}_bench.log
grep '^{' gpurun_out/${TAG}_bench.log | tail -1 > gpurun_out/${TAG}_bench.json      # the JSON line alone (copy THIS into profiles/)
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/${TAG}_prof" -o prof -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$OLDPWD/gpurun_out/${TAG}_rocprof.log" 2>&1 ); echo "rocprof rc=$?" | tee -a gpurun_out/${TAG}_rocprof.log
find gpurun_out/${TAG}_prof -name "*stats*" | head
tail -3 gpurun_out/${TAG}_pytest.log; tail -2 gpurun_out/${TAG}_smoke.log; tail -2 gpurun_out/${TAG}_bench.log
