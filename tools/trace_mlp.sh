export TMPDIR=/tmp
for v in 8192 100000000; do
  d=$PWD/gpurun_out/tr_$v
  GC_TUNE_MLP_MT2_ROWS=$v SAMPLES=1 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python3 tests/gpu_one_sample.py > /dev/null 2>&1 || exit 1
  python3 tools/trace_summary.py $d/t_kernel_trace.csv > $d/summary.txt
  rm -f $d/t_kernel_trace.csv
  echo "== MT2_ROWS=$v"; grep -E "mlp|segsum|kernel " $d/summary.txt
done
