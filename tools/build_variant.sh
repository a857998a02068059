#!/usr/bin/env bash
# Build a full libnvp_hip variant into tools/bin/libnvp_<name>.so with extra -D flags (experiments; not shipped).
# usage: build_variant.sh NAME [-DFLAG ...]     then: NVP_HIP_LIB=tools/bin/libnvp_NAME.so python bench.py ...
set -euo pipefail
cd "$(dirname "$0")/../nvp_amd/csrc"
NAME=$1; shift
OUT=../../tools/bin; mkdir -p $OUT
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -DNVP_EXPERIMENTS=1 $*"     # variants are experiments builds (every source, every switch)
objs=""; pids=()
for f in encode encode_fwd_lds encode_bwd mlp_pack mlp_fwd mlp_fwd_b3 mlp_fwd_b3r mlp_bwd mlp_bwd_b3 mlp_bwd_b3r mlp_dw mlp_dw_glds harness optim; do
  EXTRA=""; case "$f" in encode|encode_fwd_lds|encode_bwd|harness|optim) EXTRA="-ffp-contract=off";; esac
  hipcc $FL $EXTRA -c $f.hip -o $OUT/${f}__$NAME.o & pids+=($!)
  objs="$objs $OUT/${f}__$NAME.o"
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $OUT/libnvp_$NAME.so
rm -f $objs
echo "built tools/bin/libnvp_$NAME.so"
