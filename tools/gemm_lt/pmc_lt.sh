#!/bin/bash
# PMC passes over the LT GEMM harness (one shape): tools/pmc_lt.sh <shape> <outdir>
set -o pipefail
sh=${1:-1}; out=${2:-gpurun_out/pmc_lt}; mkdir -p $out; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM TCC_HIT_sum TCC_MISS_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  LT_SHAPE=$sh LT_ITERS=2 timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o p -- ./tools/bench_gemm_lt > $out/p$i.out 2> $out/p$i.err || { tail -5 $out/p$i.err; }
done
python3 - $out <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/p_counter_collection.csv", recursive=True):
  for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"]
    if "gemm" not in k: continue
    short = k.split("<")[0].split("::")[-1] + "<" + k.split("<")[1][:40] if "<" in k else k[:60]
    agg[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(agg):
  print(k)
  for c in sorted(agg[k]):
    v = agg[k][c]
    print("   %-28s n=%3d  mean %.4g" % (c, len(v), sum(v) / len(v)))
PY
