#!/usr/bin/env bash
# Build ablation variants of the bf16x3 forward kernel into tools/bin/ (experiments only; not shipped in the product).
set -euo pipefail
cd "$(dirname "$0")/../nvp_amd/csrc"
OUT=../../tools/bin; mkdir -p $OUT; rm -f $OUT/libfwd_*.so
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed"
build() { name=$1; shift; objs=""; for f in mlp_fwd mlp_fwd_b3 mlp_fwd_b3r mlp_pack; do hipcc $FL "$@" -c $f.hip -o $OUT/${f}_$name.o; objs="$objs $OUT/${f}_$name.o"; done; hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $OUT/libfwd_$name.so; rm -f $objs; }
build a_base &
build g_skeleton -DNVP_ABL_NOSPLIT -DNVP_ABL_NOSTORE -DNVP_ABL_NOSIN -DNVP_ABL_WFIXED &
build k_nosplit_nosin -DNVP_ABL_NOSPLIT -DNVP_ABL_NOSIN &
build l_nosplit_nosin_nostore -DNVP_ABL_NOSPLIT -DNVP_ABL_NOSIN -DNVP_ABL_NOSTORE &
wait
ls $OUT/libfwd_*.so
