#!/bin/bash
out=$PWD/gpurun_out/r4b
mkdir -p $out
GC_LIB_VARIANT=r256 timeout -k 10 200 python3 tests/gpu_one_degree.py > $out/att_r256.txt 2>&1 || tail -3 $out/att_r256.txt
grep -E "calls/s|attention|gemm_out" $out/att_r256.txt | sed "s/^/r256: /"
for s in 1 2 3 4; do
  GC_TUNE_ATTN_SPLITS=$s GC_LIB_VARIANT=base timeout -k 10 200 python3 tests/gpu_one_degree.py > $out/att_s$s.txt 2>&1 || tail -3 $out/att_s$s.txt
  grep -E "calls/s|attention|gemm_out|attn_comb" $out/att_s$s.txt | sed "s/^/S=$s: /"
done
