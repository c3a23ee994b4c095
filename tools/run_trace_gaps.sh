#!/bin/bash
export TMPDIR=/tmp
out=$PWD/gpurun_out/r4b
mkdir -p $out
SAMPLES=3 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/kt_gaps -o g -- python3 tests/gpu_one_sample.py > /dev/null 2> $out/kt_gaps.err || { tail -5 $out/kt_gaps.err; exit 1; }
f=$(find $out/kt_gaps -name "g_kernel_trace.csv" | head -1)
python3 tools/trace_gaps.py $f | tee $out/trace_gaps.txt
rm -rf $out/kt_gaps
