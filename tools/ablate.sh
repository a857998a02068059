#!/usr/bin/env bash
# Build ablation variants of the forward MLP kernel into tools/bin/ (not shipped in the product).
set -euo pipefail
cd "$(dirname "$0")/../nvp_amd/csrc"
OUT=../../tools/bin; mkdir -p $OUT
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed"
build() { name=$1; shift; hipcc $FL "$@" -c mlp_fwd.hip -o $OUT/fwd_$name.o && hipcc $FL "$@" -c mlp_pack.hip -o $OUT/pack_$name.o && hipcc --offload-arch=gfx950 -shared -fPIC $OUT/fwd_$name.o $OUT/pack_$name.o -o $OUT/libfwd_$name.so; }
build base &
build noload -DNVP_ABL_NOLOAD &
build noz -DNVP_ABL_NOZ &
build noload_nosin -DNVP_ABL_NOLOAD -DNVP_ABL_NOSIN &
wait
ls -la $OUT/*.so
