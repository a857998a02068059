#!/usr/bin/env bash
# Build variants of the MLP kernels into tools/bin/ (experiments only; not shipped in the product).
set -euo pipefail
cd "$(dirname "$0")/../nvp_amd/csrc"
OUT=../../tools/bin; mkdir -p $OUT; rm -f $OUT/libmlp_*.so $OUT/libfwd_*.so
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed"
build() { name=$1; shift; objs=""; for f in mlp_fwd mlp_fwd_b3 mlp_bwd mlp_bwd_b3 mlp_dw mlp_pack; do hipcc $FL "$@" -c $f.hip -o $OUT/${f}_$name.o; objs="$objs $OUT/${f}_$name.o"; done; hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $OUT/libmlp_$name.so; }
build a_split2 &
build b_share -DNVP_BWD_B3_SHARE=1 &
wait
ls $OUT/libmlp_*.so | wc -l
