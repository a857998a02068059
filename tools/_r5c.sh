cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
O=gpurun_out/r5c_fwd_ablation.txt; : > $O
for rep in 1 2; do
for v in default x2off x2_d1 x2_nostore x2_nosin x2_nomfma x2_wfixed x2off_nomfma x2off_wfixed; do
  if [ $v = default ]; then unset NVP_HIP_LIB; else export NVP_HIP_LIB=$PWD/tools/bin/libnvp_$v.so; fi
  timeout 200 python tools/fwd_time.py s 10 2>/dev/null | tail -1 | sed "s/^/$v: /" | tee -a $O
done; done
unset NVP_HIP_LIB
PASSES="sq1 sq2 sq3 sq4" bash tools/pmc.sh r5cx2 > /dev/null 2>&1
python tools/pmc_summarize.py r5cx2 s > /dev/null 2>&1; cp profiles/r5cx2_pmc_summary.txt gpurun_out/ 2>/dev/null; git checkout profiles/pmc_traffic.json 2>/dev/null
grep "mlp_fwd" gpurun_out/r5cx2_pmc_summary.txt
