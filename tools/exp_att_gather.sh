#!/bin/bash
# 1-degree attention gather experiment (round 4b): class times of the timing-only variants + TA / TCP counters of the product kernel
export TMPDIR=/tmp
out=$PWD/gpurun_out/r4b
mkdir -p $out
for v in base nok nov kcoal; do
  GC_LIB_VARIANT=$v timeout -k 10 200 python3 tests/gpu_one_degree.py > $out/att_$v.txt 2>&1 || { echo "variant $v failed"; tail -5 $out/att_$v.txt; }
  grep -E "calls/s|attention|qkv" $out/att_$v.txt | sed "s/^/$v: /"
done
rocprofv3 -L > $out/counters.txt 2>&1 || true
i=0
for set in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  ONE_DEGREE_QUICK=1 GC_LIB_VARIANT=base timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc$i -o p -- python3 tests/gpu_one_degree.py 4 > /dev/null 2> $out/pmc$i.err || { echo "pmc set $i failed"; tail -3 $out/pmc$i.err; }
done
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get("PWD") + "/gpurun_out/r4b"
for f in sorted(glob.glob(out + "/pmc*/**/p_counter_collection.csv", recursive=True)):
  acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
  for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"][:40]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); 
  for k in acc:
    if "attention" in k or "gemm_ws" in k:
      print(f.split("/")[-3] if "pmc" in f else f, k, {c: v for c, v in acc[k].items()})
PY
