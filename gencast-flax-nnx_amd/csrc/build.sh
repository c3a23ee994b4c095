#!/bin/bash
# Builds libgencast_hip.so for gfx950 (cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
  -Wall -Wno-unused-function \
  gc_kernels.hip gc_api.hip gc_graph.cpp \
  -o libgencast_hip.so "$@"
echo "built $(pwd)/libgencast_hip.so"
