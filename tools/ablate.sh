#!/usr/bin/env bash
# Build variants of the MLP kernels into tools/bin/ (experiments only; not shipped in the product).
set -euo pipefail
cd "$(dirname "$0")/../nvp_amd/csrc"
OUT=../../tools/bin; mkdir -p $OUT; rm -f $OUT/libmlp_*.so
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed"
build() { name=$1; shift; for f in mlp_fwd mlp_bwd mlp_dw mlp_pack; do hipcc $FL "$@" -c $f.hip -o $OUT/${f}_$name.o; done; hipcc --offload-arch=gfx950 -shared -fPIC $OUT/mlp_fwd_$name.o $OUT/mlp_bwd_$name.o $OUT/mlp_dw_$name.o $OUT/mlp_pack_$name.o -o $OUT/libmlp_$name.so; }
build a_base &
build b_spread1 -DNVP_STAGGER_SPREAD=1 &
build c_spread2 -DNVP_STAGGER_SPREAD=2 &
build d_sleeps8 -DNVP_STAGGER_SLEEPS=8 &
build e_sleeps16 -DNVP_STAGGER_SLEEPS=16 &
wait
ls $OUT/libmlp_*.so | wc -l
