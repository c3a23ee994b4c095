#!/bin/bash
# usage: tools/build_variant.sh <name> [-DFLAG ...]  -- EXPERIMENT builds: gc_kernels.hip (float32-feature TU only) recompiled
# with the given flags and linked with the other translation units' objects (cached under /tmp/gc_variant_obj) into
# gencast-flax-nnx_amd/csrc/variants/libgencast_hip_<name>.so.  Never the product library.
set -euo pipefail
name=$1; shift
cd "$(dirname "$0")/../gencast-flax-nnx_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
# the object cache is keyed on the hash csrc/build.sh computes over csrc/*.{hip,cpp,h,inc}: an edit of a shared include
# (argument structs in gc_kernels_decl.inc, gc_dev_common.inc) must not leave objects with the old
# layouts to be linked against new ones
TREE_HASH=$(python3 - <<'PY'
import hashlib, os
h = hashlib.sha256()
for name in sorted(os.listdir(".")):
  if name.endswith((".hip", ".cpp", ".h", ".inc")):
    h.update(name.encode())
    h.update(open(name, "rb").read())
print(h.hexdigest()[:16])
PY
)
OBJ=/tmp/gc_variant_obj/$TREE_HASH
mkdir -p "$OBJ" variants
SRC_HASH=variant-$name
BASE=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DGC_SOURCE_HASH="\"$SRC_HASH\"")
pids=()
for src in gc_api.hip gc_noise.hip gc_graph.cpp; do
  if [ ! -f "$OBJ/${src%.*}.o" ] || [ "$src" -nt "$OBJ/${src%.*}.o" ]; then
    "$HIPCC" "${BASE[@]}" -c "$src" -o "$OBJ/${src%.*}.o" & pids+=($!)
  fi
done
if [ ! -f "$OBJ/gc_kernels_a16.o" ] || [ gc_kernels.hip -nt "$OBJ/gc_kernels_a16.o" ]; then
  "$HIPCC" "${BASE[@]}" -DGC_TU_A16 -c gc_kernels.hip -o "$OBJ/gc_kernels_a16.o" & pids+=($!)
fi
"$HIPCC" "${BASE[@]}" "$@" -c gc_kernels.hip -o "$OBJ/gc_kernels_$name.o" & pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -fPIC -shared "$OBJ/gc_kernels_$name.o" "$OBJ"/gc_kernels_a16.o "$OBJ"/gc_api.o "$OBJ"/gc_noise.o "$OBJ"/gc_graph.o -o "variants/libgencast_hip_$name.so"
echo "built variants/libgencast_hip_$name.so"
