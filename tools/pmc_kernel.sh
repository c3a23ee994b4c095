#!/bin/bash
# usage: tools/pmc_kernel.sh <tag> "<COUNTERS set 1>" ["<set 2>" ...]  -- per-kernel PMC averages over one sample
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_$tag
mkdir -p $out
i=0
for set in "$@"; do
  i=$((i+1))
  SAMPLES=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/s$i -o p -- python3 tests/gpu_one_sample.py > /dev/null 2> $out/s$i.err || { tail -5 $out/s$i.err; exit 1; }
done
python3 tools/pmc_traffic.py $out/$tag $out/s*/p_counter_collection.csv > $out/summary.txt
rm -rf $out/s*/
