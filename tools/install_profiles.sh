#!/bin/bash
# Copies what tools/profile_round.sh collected (gpurun_out/prof_<tag>/) into profiles/ under the round's names.
# Run on the GPU box between the profile and bench.py (so the bench line can quote the fresh figures) and again
# in the build container after gpurun has merged gpurun_out/ back.      usage: tools/install_profiles.sh r03
set -e
tag=${1:-r03}
src=gpurun_out/prof_$tag
cp $src/kernel_trace_summary.txt profiles/${tag}_kernel_trace_summary.txt
cp $src/one_degree_kernel_trace_summary.txt profiles/${tag}_1deg_kernel_trace_summary.txt
cp $src/kt/${tag}_kernel_stats.csv profiles/${tag}_kernel_stats.csv
cp $src/kt1/${tag}_1deg_kernel_stats.csv profiles/${tag}_1deg_kernel_stats.csv
cp $src/bench_under_rocprof.json profiles/${tag}_bench_under_rocprof.json
cp $src/${tag}_pmc_per_kernel.json $src/${tag}_fp16_features_pmc_per_kernel.json $src/traffic.json $src/profile_meta.json profiles/
cp $src/${tag}_one_degree_pmc_per_kernel.json $src/${tag}_one_degree_fp16_features_pmc_per_kernel.json profiles/ 2>/dev/null || true
echo "profiles/ <- $src ($(cat $src/profile_meta.json))"
