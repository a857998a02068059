#!/usr/bin/env bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
NVP_PARITY_REPORT=1 timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/r3b_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3b_pytest.log
tail -5 gpurun_out/r3b_pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r3b_bench.json 2> gpurun_out/r3b_bench.err; tail -c 600 gpurun_out/r3b_bench.json
python bench.py --mode eval --steps 8 --warmup 2 > gpurun_out/r3b_bench_eval.json 2> gpurun_out/r3b_bench_eval.err; cat gpurun_out/r3b_bench_eval.json | head -c 3000
NVP_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 3 --warmup 1 --dp auto > gpurun_out/r3b_bench_gloo2.json 2> gpurun_out/r3b_bench_gloo2.err; echo "gloo2 rc=$?"; tail -c 400 gpurun_out/r3b_bench_gloo2.json
for A in 1.3 1.6; do
  python -m nvp_amd.train --video natural --natural-alpha $A --natural-grain 1.0 --seconds 25 --report-every 1000 --log gpurun_out/r3b_nat_a$A.jsonl > gpurun_out/r3b_nat_a$A.log 2>&1; tail -1 gpurun_out/r3b_nat_a$A.jsonl
done
