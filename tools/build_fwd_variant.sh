#!/usr/bin/env bash
# Variant of the FORWARD kernels only: mlp_fwd_b3.hip recompiled with extra -D flags and linked with the product library's other
# objects (nvp_amd/csrc/obj/*.o: run nvp_amd/csrc/build.sh first) -> tools/bin/libnvp_<name>.so.   usage: build_fwd_variant.sh NAME [-DFLAG ...]
set -euo pipefail
cd "$(dirname "$0")/../nvp_amd/csrc"
NAME=$1; shift
OUT=../../tools/bin; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -Wno-constant-logical-operand "$@" -c mlp_fwd_b3.hip -o $OUT/mlp_fwd_b3__$NAME.o
objs=$(ls obj/*.o | grep -v "obj/mlp_fwd_b3.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $OUT/mlp_fwd_b3__$NAME.o -o $OUT/libnvp_$NAME.so
rm -f $OUT/mlp_fwd_b3__$NAME.o
echo "built tools/bin/libnvp_$NAME.so"
