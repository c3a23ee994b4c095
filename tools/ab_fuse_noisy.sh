#!/bin/bash
out=$PWD/gpurun_out/r4b
mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "sampl or churn or rollout or graph or smoke or ensemble or member" > $out/noisy_tests.log 2>&1 || { tail -20 $out/noisy_tests.log; exit 1; }
tail -2 $out/noisy_tests.log
for p in 1 0 1 0 1 0; do
  GC_TUNE_FUSE_NOISY=$p timeout -k 10 200 python3 tests/gpu_class_timing.py > $out/noisy_nano_$p.txt 2>&1 || tail -3 $out/noisy_nano_$p.txt
  grep -E "calls/s|gc_pack" $out/noisy_nano_$p.txt | sed "s/^/nano fuse_noisy=$p: /"
done
