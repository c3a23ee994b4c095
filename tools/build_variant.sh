#!/bin/bash
# usage: tools/build_variant.sh <name> [-DFLAG ...]  -- EXPERIMENT builds: gc_kernels.hip (float32-feature TU only) recompiled
# with the given flags and linked with the other translation units' objects (cached under /tmp/gc_variant_obj) into
# gencast-flax-nnx_amd/csrc/variants/libgencast_hip_<name>.so.  Never the product library.
set -euo pipefail
name=$1; shift
cd "$(dirname "$0")/../gencast-flax-nnx_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OBJ=/tmp/gc_variant_obj
mkdir -p "$OBJ" variants
SRC_HASH=variant-$name
BASE=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DGC_SOURCE_HASH="\"$SRC_HASH\"")
pids=()
for src in gc_gemm_lt.hip gc_api.hip gc_noise.hip gc_graph.cpp; do
  if [ ! -f "$OBJ/${src%.*}.o" ] || [ "$src" -nt "$OBJ/${src%.*}.o" ]; then
    "$HIPCC" "${BASE[@]}" -c "$src" -o "$OBJ/${src%.*}.o" & pids+=($!)
  fi
done
if [ ! -f "$OBJ/gc_kernels_a16.o" ] || [ gc_kernels.hip -nt "$OBJ/gc_kernels_a16.o" ]; then
  "$HIPCC" "${BASE[@]}" -DGC_TU_A16 -c gc_kernels.hip -o "$OBJ/gc_kernels_a16.o" & pids+=($!)
fi
"$HIPCC" "${BASE[@]}" "$@" -c gc_kernels.hip -o "$OBJ/gc_kernels_$name.o" & pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -fPIC -shared "$OBJ/gc_kernels_$name.o" "$OBJ"/gc_kernels_a16.o "$OBJ"/gc_gemm_lt.o "$OBJ"/gc_api.o "$OBJ"/gc_noise.o "$OBJ"/gc_graph.o -o "variants/libgencast_hip_$name.so"
echo "built variants/libgencast_hip_$name.so"
