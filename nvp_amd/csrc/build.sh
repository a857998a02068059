#!/usr/bin/env bash
# Build libnvp_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
# A failed compile aborts the build: its old object is removed first and every job's exit status is
# checked, so the link can never pick up a stale object (bare `wait` returns 0 whatever the jobs did).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed ${NVP_EXTRA_FLAGS:-}"
OBJS=()
PIDS=()
NAMES=()
for f in encode encode_fwd_lds encode_bwd mlp_pack mlp_fwd mlp_fwd_b3 mlp_fwd_b3r mlp_bwd mlp_bwd_b3 mlp_bwd_b3r mlp_dw harness optim; do
  [ -f "$f.hip" ] || continue
  if [ ! -f "$f.o" ] || [ "$f.hip" -nt "$f.o" ] || [ -n "$(find . -maxdepth 1 -name '*.h' -newer "$f.o" 2>/dev/null)" ] || [ ../../include/nvp_hip.h -nt "$f.o" ] || [ -n "${NVP_REBUILD:-}" ]; then
    EXTRA=""
    case "$f" in encode|encode_fwd_lds|encode_bwd|harness|optim) EXTRA="-ffp-contract=off";; esac   # separately rounded mul/add (index parity)
    rm -f "$f.o"
    "$HIPCC" $FLAGS $EXTRA -c "$f.hip" -o "$f.o" &
    PIDS+=($!)
    NAMES+=("$f")
  fi
  OBJS+=("$f.o")
done
rc=0
for i in "${!PIDS[@]}"; do
  if ! wait "${PIDS[$i]}"; then echo "build.sh: compiling ${NAMES[$i]}.hip FAILED" >&2; rm -f "${NAMES[$i]}.o"; rc=1; fi
done
if [ $rc -ne 0 ]; then rm -f libnvp_hip.so; exit 1; fi
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${OBJS[@]}" -o libnvp_hip.so
echo "built $(pwd)/libnvp_hip.so"
