#!/usr/bin/env bash
# Build libnvp_hip variants of the scatter stage into tools/bin/ (experiments only): current tree vs the previous commit's permute.
set -euo pipefail
cd "$(dirname "$0")/../nvp_amd/csrc"
OUT=../../tools/bin; mkdir -p $OUT; rm -f $OUT/libnvp_*.so
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -ffp-contract=off"
OTHERS="encode.o mlp_pack.o mlp_fwd.o mlp_fwd_b3.o mlp_bwd.o mlp_bwd_b3.o mlp_dw.o harness.o optim.o"
hipcc $FL -c encode_bwd.hip -o $OUT/encbwd_a_new.o && hipcc --offload-arch=gfx950 -shared -fPIC $OUT/encbwd_a_new.o $OTHERS -o $OUT/libnvp_a_new.so
git show HEAD:nvp_amd/csrc/encode_bwd.hip > /tmp/encode_bwd_old.hip
hipcc $FL -I. -I../../include -c /tmp/encode_bwd_old.hip -o $OUT/encbwd_b_old.o && hipcc --offload-arch=gfx950 -shared -fPIC $OUT/encbwd_b_old.o $OTHERS -o $OUT/libnvp_b_old.so
ls $OUT/libnvp_*.so
