#!/usr/bin/env bash
# round-4 second check: fused dense AdamW + reworked long test, bench, kernel trace incl. the reference-surface pass, PMC diagnosis of the
# forward kernel (instruction fetch, VMEM / LDS latency levels, L2 hit rates), memory-side cache probe
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
T=${1:-r4b}
NVP_PARITY_REPORT=1 timeout 1500 python -m pytest tests/test_gpu_parity.py::test_early_grid_update_equals_the_in_order_optimizer_step tests/test_gpu_parity.py::test_psnr_after_1000_steps_matches_oracle tests/test_gpu_parity.py::test_kernel_variants_are_bit_identical tests/test_gpu_parity.py::test_forwards_whose_graph_is_dropped_do_not_disturb_training -q --timeout 1200 --durations=5 > gpurun_out/${T}_new.log 2>&1; echo "new rc=$?"; tail -12 gpurun_out/${T}_new.log | cut -c1-250
grep psnr_equal_steps_long gpurun_out/parity_report.jsonl | cut -c1-700
for rep in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${T}_bench_s$rep.json 2> gpurun_out/${T}_bench_s$rep.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open('gpurun_out/${T}_bench_s$rep.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['kernels_ms'], 'iso', d['isolated']['ms_per_step'], d['isolated']['kernels_ms'], 'ref', d['reference_surface']['ms_per_step'])
PY
done
NVP_FUSED_DENSE_ADAMW=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-isolate --no-reference-surface 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dense-adamw-off', d['ms_per_step'], d['kernels_ms'])"
# kernel trace incl. the reference-surface pass
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/${T}_refprof" -o prof --output-format csv -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --prewarm 0 --no-cpu-baseline --no-isolate > "$OLDPWD/gpurun_out/${T}_refprof.log" 2>&1 )
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/${T}_refprof/**/*kernel_stats.csv', recursive=True)
rows=list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
out=open('gpurun_out/${T}_refprof_stats.txt','w')
for r in rows[:45]:
    line=f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} total_ms {float(r['TotalDurationNs'])/1e6:9.3f} avg_us {float(r['AverageNs'])/1e3:9.1f}"
    print(line); out.write(line+'\n')
PY
PASSES="sq3 sq4 tcc" bash tools/pmc.sh ${T}s > gpurun_out/${T}_pmc.log 2>&1; python tools/pmc_summarize.py ${T}s s > /dev/null 2>&1; grep -E "mlp_fwd_b3|mlp_bwd_b3r|mlp_dw_kernel|band_kernel" profiles/${T}s_pmc_summary.txt | cut -c1-600
timeout 300 tools/bin/mall_probe > gpurun_out/${T}_mall_probe.txt 2>&1; cat gpurun_out/${T}_mall_probe.txt
