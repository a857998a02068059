#!/usr/bin/env bash
# PMC passes (counters only with --kernel-trace; each group in its own run). Output -> gpurun_out/<tag>_pmc*/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-pmc}
BENCH_ARGS=${BENCH_ARGS:-}            # e.g. BENCH_ARGS="--config l" bash tools/pmc.sh r03l
PASSES=${PASSES:-"sq1 sq2 fetch write"}
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]+|GRBM_[A-Z_]+|TCC_[A-Z_0-9]+|FETCH_SIZE|WRITE_SIZE|MfmaUtil|VALUBusy|OccupancyPercent)\b" | sort -u > gpurun_out/${TAG}_counters.txt
grep -cE "." gpurun_out/${TAG}_counters.txt
run() { # name counters...
  local name=$1; shift
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc "$@" -d "$OLDPWD/gpurun_out/${TAG}_$name" -o pmc --output-format csv -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --prewarm 0 --no-cpu-baseline --no-arithmetic-check --no-isolate --no-reference-surface --no-other-configs --no-dp-floor --no-confirm $BENCH_ARGS > "$OLDPWD/gpurun_out/${TAG}_$name.log" 2>&1 ); echo "$name rc=$?"
}
for P in $PASSES; do
  case $P in
    sq1) run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY;;
    sq2) run sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE;;
    sq3) run sq3 SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVE_CYCLES;;
    sq4) run sq4 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS;;
    tcc) run tcc TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_READ;;
    fetch) run fetch FETCH_SIZE;;
    write) run write WRITE_SIZE;;
  esac
done
ls gpurun_out/${TAG}_sq1 | head
