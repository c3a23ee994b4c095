#!/bin/bash
# A/B sweep of launch-geometry switches at the 1-degree full-width size (tests/gpu_one_degree.py).
for cfg in "" "GC_TUNE_ATTN_SPLITS=2" "GC_TUNE_ATTN_SPLITS=3" "GC_TUNE_WS_MT=1" "GC_TUNE_FFW_FUSED=2" "GC_TUNE_MLP_WS512=1"; do
  echo "== $cfg"
  env $cfg python tests/gpu_one_degree.py 2>&1 | grep -E "1deg:|attention|ffw|gc_mlp|qkv|gemm_out"
done
