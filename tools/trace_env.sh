#!/bin/bash
# usage: tools/trace_env.sh <grep pattern> "ENV=val ..." ["ENV=val ..."]: kernel-trace summary of one sample per env set
pat=$1; shift
export TMPDIR=/tmp
i=0
for e in "$@"; do
  i=$((i+1)); d=$PWD/gpurun_out/tre_$i
  env $e SAMPLES=1 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python3 tests/gpu_one_sample.py > /dev/null 2>&1 || exit 1
  python3 tools/trace_summary.py $d/t_kernel_trace.csv > $d/summary.txt; rm -f $d/t_kernel_trace.csv
  echo "== $e"; grep -E "$pat" $d/summary.txt
done
