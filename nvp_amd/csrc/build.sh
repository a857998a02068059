#!/usr/bin/env bash
# Build libnvp_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed ${NVP_EXTRA_FLAGS:-}"
OBJS=()
for f in encode encode_bwd mlp_pack mlp_fwd mlp_fwd_b3 mlp_bwd mlp_bwd_b3 mlp_dw harness optim; do
  if [ ! -f "$f.o" ] || [ "$f.hip" -nt "$f.o" ] || [ -n "$(find . -maxdepth 1 -name '*.h' -newer "$f.o" 2>/dev/null)" ] || [ ../../include/nvp_hip.h -nt "$f.o" ]; then
    EXTRA=""
    case "$f" in encode|encode_bwd|harness|optim) EXTRA="-ffp-contract=off";; esac   # separately rounded mul/add (index parity)
    "$HIPCC" $FLAGS $EXTRA -c "$f.hip" -o "$f.o" &
  fi
  OBJS+=("$f.o")
done
wait
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${OBJS[@]}" -o libnvp_hip.so
echo "built $(pwd)/libnvp_hip.so"
