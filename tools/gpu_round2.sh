#!/usr/bin/env bash
# round-2 evidence run: bench lines for configs[1..3], rocprof kernel stats, DP code paths with one rank, PSNR bisect
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
TAG=${1:-r2}
export TMPDIR=/tmp
stage() { local name=$1 to=$2; shift 2; echo "=== $name"; ( timeout $to "$@" ) > gpurun_out/${TAG}_$name.log 2>&1; echo "$name rc=$?"; }
stage bench_s 600 python bench.py --steps 20 --warmup 5
grep '^{' gpurun_out/${TAG}_bench_s.log | tail -1 > gpurun_out/${TAG}_bench_s.json
stage bench_l 600 python bench.py --steps 10 --warmup 3 --config l --no-cpu-baseline
grep '^{' gpurun_out/${TAG}_bench_l.log | tail -1 > gpurun_out/${TAG}_bench_l.json
stage bench_4k 600 python bench.py --steps 10 --warmup 3 --config 4k --no-cpu-baseline
grep '^{' gpurun_out/${TAG}_bench_4k.log | tail -1 > gpurun_out/${TAG}_bench_4k.json
for C in s l 4k; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/${TAG}_prof_$C" -o prof --output-format csv -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --config $C > "$OLDPWD/gpurun_out/${TAG}_rocprof_$C.log" 2>&1 )
  f=$(find gpurun_out/${TAG}_prof_$C -name "*kernel_stats*" | head -1)
  python - "$f" "$C" > gpurun_out/${TAG}_kernel_stats_$C.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print(f"# rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --config {sys.argv[2]}   (7 steps captured; microseconds)")
print(f"{'calls':>6} {'total_us':>12} {'avg_us':>10} {'pct':>6}  kernel")
for r in rows[:40]:
    print(f"{int(r['Calls']):>6} {float(r['TotalDurationNs'])/1e3:>12.1f} {float(r['AverageNs'])/1e3:>10.1f} {float(r['Percentage']):>6.2f}  {r['Name'][:160]}")
PY
done
for M in auto sharded a2a replicated; do
  stage dp1_$M 600 env NVP_DP_FORCE_COLLECTIVES=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --dp $M
  grep '^{' gpurun_out/${TAG}_dp1_$M.log | tail -1 > gpurun_out/${TAG}_dp1_$M.json
done
[ -n "${SKIP_BISECT:-}" ] || stage psnr_bisect 1500 python tools/psnr_bisect.py
for f in gpurun_out/${TAG}_bench_*.json gpurun_out/${TAG}_dp1_*.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f', d['ms_per_step'], d['value'], d.get('dp',{}).get('mode'), d.get('dp',{}).get('autotune_ms_per_step'), d.get('dp',{}).get('post_backward_ms_per_rank'))"; done
[ -n "${SKIP_BISECT:-}" ] || tail -8 gpurun_out/${TAG}_psnr_bisect.log
