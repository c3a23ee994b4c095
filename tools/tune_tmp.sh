for s in 2 3 4 5; do echo "ATTN_SPLITS=$s"; GC_TUNE_ATTN_SPLITS=$s python tests/gpu_class_timing.py 2>&1 | grep -E "sample:|attention |gemm_out"; done
