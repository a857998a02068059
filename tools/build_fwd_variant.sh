#!/usr/bin/env bash
# Variant of ONE source of the product library: SRC.hip (default mlp_fwd_b3) recompiled with extra -D flags and linked with the product
# library's other objects (nvp_amd/csrc/obj/*.o: run nvp_amd/csrc/build.sh first) -> tools/bin/libnvp_<name>.so.
# usage: [SRC=encode_bwd] build_fwd_variant.sh NAME [-DFLAG ...]      then: NVP_HIP_LIB=tools/bin/libnvp_NAME.so python ...
set -euo pipefail
cd "$(dirname "$0")/../nvp_amd/csrc"
NAME=$1; shift
SRC=${SRC:-mlp_fwd_b3}
mkdir -p ../../tools/bin
OBJ="../../tools/bin/${SRC}__${NAME}.variant.o"
EXTRA=""; case "$SRC" in encode|encode_bwd|harness|optim) EXTRA="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -Wno-constant-logical-operand $EXTRA "$@" -c "$SRC.hip" -o "$OBJ"
objs=$(ls obj/*.o | grep -v "obj/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs "$OBJ" -o "../../tools/bin/libnvp_$NAME.so"
echo "built tools/bin/libnvp_$NAME.so"
