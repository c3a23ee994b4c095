#!/bin/bash
# Builds libgencast_hip.so for gfx950 (cross-compiles without a GPU).  The translation
# units are compiled side by side, then linked; objects go to a scratch directory.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
# `build.sh stamps` builds the DIAGNOSTIC library libgencast_hip_stamps.so (-DGC_STAMPS: in-kernel s_memtime
# stamps for tools/stamp_attention.py); the product library is never built with it.
OUT=libgencast_hip.so
if [ "${1:-}" = "stamps" ]; then shift; set -- -DGC_STAMPS "$@"; OUT=libgencast_hip_stamps.so; fi
# the hash bench.py's source_hash() computes over these sources, compiled into gc_build_info(): a library that was
# built from OTHER sources than the tree it sits in (an edit or a `git checkout` without a rebuild) is then caught by
# tests/test_abi.py and by bench.py instead of being measured under the wrong name
SRC_HASH=$(python3 - <<'PY'
import hashlib, os
h = hashlib.sha256()
for name in sorted(os.listdir(".")):
  if name.endswith((".hip", ".cpp", ".h", ".inc")):
    h.update(name.encode())
    h.update(open(name, "rb").read())
print(h.hexdigest()[:16])
PY
)
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DGC_SOURCE_HASH="\"$SRC_HASH\"" "$@")
OBJ=$(mktemp -d)
trap 'rm -rf "$OBJ"' EXIT
pids=()
for src in gc_kernels.hip gc_api.hip gc_noise.hip gc_graph.cpp; do
  "$HIPCC" "${FLAGS[@]}" -c "$src" -o "$OBJ/${src%.*}.o" &
  pids+=($!)
done
# the same kernels again as namespace gc_a16: 2-MFMA variants for exact-fp16 activations (gc_kernels.hip, top)
"$HIPCC" "${FLAGS[@]}" -DGC_TU_A16 -c gc_kernels.hip -o "$OBJ/gc_kernels_a16.o" &
pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -fPIC -shared "$OBJ"/gc_kernels.o "$OBJ"/gc_kernels_a16.o "$OBJ"/gc_api.o "$OBJ"/gc_noise.o "$OBJ"/gc_graph.o -o "$OUT"
echo "built $(pwd)/$OUT (sources $SRC_HASH)"
