#!/usr/bin/env bash
# rocprofv3 kernel stats of a short bench run; usage: [BENCH_ARGS="--config l"] gpu_prof.sh TAG [env assignments...]
# (isolated stage durations: gpu_prof.sh TAG NVP_EARLY_ADAMW=0 NVP_SCATTER_PRESORT=0 NVP_SAMPLER_PREFETCH=0)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
( cd /tmp && env "$@" timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/${TAG}_prof" -o prof --output-format csv -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --prewarm 0 --no-cpu-baseline --no-arithmetic-check --no-isolate --no-reference-surface --no-other-configs --no-dp-floor --no-confirm ${BENCH_ARGS:-} > "$OLDPWD/gpurun_out/${TAG}_rocprof.log" 2>&1 )
echo "rocprof rc=$?"
f=$(find gpurun_out/${TAG}_prof -name "*kernel_stats*" | head -1)
python - "$f" <<'PY' | tee gpurun_out/${TAG}_kernel_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print(f"{'calls':>6} {'total_us':>12} {'avg_us':>10} {'pct':>6}  kernel")
for r in rows[:28]:
    print(f"{int(r['Calls']):>6} {float(r['TotalDurationNs'])/1e3:>12.1f} {float(r['AverageNs'])/1e3:>10.1f} {float(r['Percentage']):>6.2f}  {r['Name'][:150]}")
PY
