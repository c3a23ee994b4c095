#!/bin/bash
# usage: tools/ab_f16_1deg.sh "ENV=val ..." ...: class times of the 1-degree workload per feature mode for each environment
for e in "$@"; do
  echo "== $e"
  env $e python tools/class_times_by_feature_mode.py one_degree 2>&1 | grep -v amdgpu.ids
done
