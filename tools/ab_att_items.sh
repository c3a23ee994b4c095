#!/bin/bash
# A/B of the attention work-item list at the 1-degree size (GC_TUNE_ATTN_ITEMS=0|1): parity tests, then class times in both feature modes and at k_hop 16
out=$PWD/gpurun_out/r4b
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_api.py -m gpu -x -q -k "one_degree or 1deg or khop16 or rollout" > $out/items_tests.log 2>&1 || { tail -20 $out/items_tests.log; exit 1; }
tail -2 $out/items_tests.log
for f in f32 f16; do for p in 0 1 0 1; do
  GC_FEATURES=$f GC_TUNE_ATTN_ITEMS=$p timeout -k 10 200 python3 tests/gpu_one_degree.py > $out/items_1deg_${f}_$p.txt 2>&1 || tail -3 $out/items_1deg_${f}_$p.txt
  grep -E "calls/s|attention|gemm_out" $out/items_1deg_${f}_$p.txt | tr '\n' ' ' | sed "s/^/1deg $f items=$p: /"; echo
done; done
for p in 0 1; do
  GC_TUNE_ATTN_ITEMS=$p timeout -k 10 200 python3 tests/gpu_one_degree.py 16 16 > $out/items_k16_$p.txt 2>&1 || tail -3 $out/items_k16_$p.txt
  grep -E "calls/s|attention|gemm_out" $out/items_k16_$p.txt | tr '\n' ' ' | sed "s/^/k_hop16 items=$p: /"; echo
done
