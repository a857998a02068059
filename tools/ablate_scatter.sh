#!/usr/bin/env bash
# Build libnvp_hip variants with different scatter table geometries into tools/bin/ (experiments only).
set -euo pipefail
cd "$(dirname "$0")/../nvp_amd/csrc"
OUT=../../tools/bin; mkdir -p $OUT
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -ffp-contract=off"
build() { name=$1; shift; hipcc $FL "$@" -c encode_bwd.hip -o $OUT/encbwd_$name.o && hipcc --offload-arch=gfx950 -shared -fPIC $OUT/encbwd_$name.o encode.o mlp_pack.o mlp_fwd.o mlp_bwd.o mlp_dw.o harness.o -o $OUT/libnvp_$name.so; }
build c_512_6k_s3k -DNVP_BAND_THREADS=512 -DNVP_BAND_ENTRIES=6000 -DNVP_SPARSE_ENTRIES=3000 &
build c_512_4k5_s3k -DNVP_BAND_THREADS=512 -DNVP_BAND_ENTRIES=4500 -DNVP_SPARSE_ENTRIES=3000 &
build c_512_6k_s2k -DNVP_BAND_THREADS=512 -DNVP_BAND_ENTRIES=6000 -DNVP_SPARSE_ENTRIES=2000 &
wait
build c_256_6k_s3k -DNVP_BAND_THREADS=256 -DNVP_BAND_ENTRIES=6000 -DNVP_SPARSE_ENTRIES=3000 &
build c_512_7k5_s3k -DNVP_BAND_THREADS=512 -DNVP_BAND_ENTRIES=7500 -DNVP_SPARSE_ENTRIES=3000 &
wait
ls $OUT/libnvp_*.so
