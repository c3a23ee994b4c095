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
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@")
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
echo "built $(pwd)/$OUT"
