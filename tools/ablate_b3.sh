#!/usr/bin/env bash
# Build ablation variants of the bf16x3 forward kernel into tools/bin/ (experiments only; not shipped in the product).
set -euo pipefail
cd "$(dirname "$0")/../nvp_amd/csrc"
OUT=../../tools/bin; mkdir -p $OUT; rm -f $OUT/libfwd_*.so
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed"
build() { name=$1; shift; objs=""; for f in mlp_fwd mlp_fwd_b3 mlp_fwd_b3r mlp_pack; do hipcc $FL "$@" -c $f.hip -o $OUT/${f}_$name.o; objs="$objs $OUT/${f}_$name.o"; done; hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $OUT/libfwd_$name.so; rm -f $objs; }
build a_base &
build m_swp -DNVP_B3_SWP=1 &
wait
ls $OUT/libfwd_*.so
