#!/bin/bash
# 1-degree attention ablations (round 4, second session; profiles/r04_attention_gather_ablation.txt).
#   in the build container:  tools/exp_att_gather.sh build     -> csrc/variants/libgencast_hip_{base,nok,nov,kcoal,r256}.so
#   on the GPU box:          tools/exp_att_gather.sh            -> class times per variant, key splits 1..4, TA / TCP counters
# The nok / nov / kcoal / r256 variants compute WRONG values by construction (timing only).
if [ "${1:-}" = "build" ]; then
  set -e
  tools/build_variant.sh base
  tools/build_variant.sh nok -DGC_EXP_ATT_NOK
  tools/build_variant.sh nov -DGC_EXP_ATT_NOV
  tools/build_variant.sh kcoal -DGC_EXP_ATT_KCOAL
  tools/build_variant.sh r256 -DGC_EXP_ATT_256
  exit 0
fi
export TMPDIR=/tmp
out=$PWD/gpurun_out/att_ablation
mkdir -p $out
for v in base nok nov kcoal r256; do
  GC_ALLOW_EXPERIMENT_LIB=1 GC_LIB_VARIANT=$v timeout -k 10 200 python3 tests/gpu_one_degree.py > $out/att_$v.txt 2>&1 || { echo "variant $v failed"; tail -5 $out/att_$v.txt; }
  grep -E "calls/s|attention|qkv" $out/att_$v.txt | sed "s/^/$v: /"
done
for s in 1 2 3 4; do
  GC_TUNE_ATTN_SPLITS=$s GC_ALLOW_EXPERIMENT_LIB=1 GC_LIB_VARIANT=base timeout -k 10 200 python3 tests/gpu_one_degree.py > $out/att_s$s.txt 2>&1 || tail -3 $out/att_s$s.txt
  grep -E "calls/s|attention|gemm_out" $out/att_s$s.txt | sed "s/^/S=$s: /"
done
i=0
for set in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  ONE_DEGREE_QUICK=1 GC_ALLOW_EXPERIMENT_LIB=1 GC_LIB_VARIANT=base timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc$i -o p -- python3 tests/gpu_one_degree.py 4 > /dev/null 2> $out/pmc$i.err || { echo "pmc set $i failed"; tail -3 $out/pmc$i.err; }
done
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get("PWD") + "/gpurun_out/att_ablation"
for f in sorted(glob.glob(out + "/pmc*/**/p_counter_collection.csv", recursive=True)):
  acc = collections.defaultdict(lambda: collections.defaultdict(float))
  for row in csv.DictReader(open(f)):
    acc[row["Kernel_Name"][:40]][row["Counter_Name"]] += float(row["Counter_Value"])
  for k in acc:
    if "attention" in k:
      print(k, dict(acc[k]))
PY
