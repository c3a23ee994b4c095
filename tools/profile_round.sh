#!/bin/bash
# Collects the per-round evidence on a GPU box: kernel-trace stats of bench.py, then separate
# --pmc passes (HBM traffic, MFMA busy) over one 39-call sample.  usage: tools/profile_round.sh r01
set -o pipefail
tag=${1:-r02}
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
# refuse to profile a library that was not built from this tree's sources (csrc/build.sh stamps the hash)
python3 -c "import sys; sys.path.insert(0,'.'); import bench; sys.exit(0 if bench.library_source_hash() == bench.source_hash() else 'profile_round: libgencast_hip.so is stale, rebuild first')" || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o $tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --rollout-steps 0 > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err || exit 1
for set in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  d=$out/pmc_$(echo $set | cut -c1-12 | tr ' ' '_')
  SAMPLES=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -o p -- python3 tests/gpu_one_sample.py > /dev/null 2> $d.err || exit 1
  rm -f $d/p_kernel_trace.csv $d/p_agent_info.csv
done
# the same PMC passes in fp16-feature mode (physical fp16 activation storage: BASELINE configs[4]'s arithmetic)
for set in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  d=$out/pmc16_$(echo $set | cut -c1-12 | tr ' ' '_')
  GC_FEATURES=f16 SAMPLES=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -o p -- python3 tests/gpu_one_sample.py > /dev/null 2> $d.err || exit 1
  rm -f $d/p_kernel_trace.csv $d/p_agent_info.csv
done
# BASELINE configs[3] (1 deg, full width): kernel stats of a few denoiser calls + one class-profiled pass
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt1 -o ${tag}_1deg -- python3 tests/gpu_one_degree.py > $out/one_degree_under_rocprof.txt 2> $out/one_degree_under_rocprof.err || exit 1
python3 tools/trace_summary.py $out/kt1/${tag}_1deg_kernel_trace.csv > $out/one_degree_kernel_trace_summary.txt
# ... and its own PMC passes (the 1-degree roofline had durations but no counters): memory-side bytes and
# MFMA-busy per kernel at the 1-degree sizes, both feature modes
for mode in f32 f16; do
  for set in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    d=$out/pmc1deg_${mode}_$(echo $set | cut -c1-12 | tr ' ' '_')
    GC_FEATURES=$mode ONE_DEGREE_QUICK=1 timeout -k 10 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -o p -- python3 tests/gpu_one_degree.py > /dev/null 2> $d.err || exit 1
    rm -f $d/p_kernel_trace.csv $d/p_agent_info.csv
  done
done

rm -f $out/kt1/${tag}_1deg_kernel_trace.csv
python3 tools/pmc_traffic.py $out/$tag $out/pmc_*/p_counter_collection.csv > $out/pmc_summary.txt
python3 tools/pmc_traffic.py --mode fp16_features $out/$tag $out/pmc16_*/p_counter_collection.csv > $out/pmc_summary_fp16_features.txt
# (after the two calls above: the base call rewrites traffic.json, the --mode calls add to it)
python3 tools/pmc_traffic.py --mode one_degree $out/$tag $out/pmc1deg_f32_*/p_counter_collection.csv > $out/pmc_summary_one_degree.txt
python3 tools/pmc_traffic.py --mode one_degree_fp16_features $out/$tag $out/pmc1deg_f16_*/p_counter_collection.csv > $out/pmc_summary_one_degree_fp16_features.txt
rm -f $out/pmc1deg_*/p_counter_collection.csv
python3 -c "import json,sys; sys.path.insert(0,'.'); import bench; json.dump({'tag':'$tag','source_hash':bench.source_hash()}, open('$out/profile_meta.json','w'))"
python3 tools/trace_summary.py $out/kt/${tag}_kernel_trace.csv > $out/kernel_trace_summary.txt
rm -f $out/kt/${tag}_kernel_trace.csv $out/pmc_*/p_counter_collection.csv $out/pmc16_*/p_counter_collection.csv
ls -la $out $out/kt
