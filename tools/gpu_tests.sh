#!/usr/bin/env bash
# the -m gpu suite with per-test timeouts, durations and the parity report; usage: gpu_tests.sh TAG [pytest args...]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
TAG=${1:-t}; shift || true
rm -f gpurun_out/parity_report.jsonl
NVP_PARITY_REPORT=1 timeout ${SUITE_TIMEOUT:-1700} python -m pytest tests -m gpu -q --tb=short --timeout ${TEST_TIMEOUT:-400} --durations=30 "$@" > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/${TAG}_pytest.log
tail -60 gpurun_out/${TAG}_pytest.log | cut -c1-220
