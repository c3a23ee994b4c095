#!/bin/bash
# Evidence run of the large-tile GEMM harness (tools/bench_gemm_lt, built as the header of tools/gemm_lt/bench_gemm_lt.cpp says): every kernel generation
# next to the weight-streaming kernels on one box, the stamped generation-2 run, the ablations of generation 3
# (LT_DBG bits: 2 = no copies, 4 = no fragment reads / MFMAs, 8 = epilogue arithmetic without its stores, 16 = write-through
# stores) and constant operands.      usage: tools/lt_evidence.sh > profiles/rNN_lt_gemm_harness.txt
set -o pipefail
B=./tools/bench_gemm_lt
run() { echo "== $*"; env "$@" LT_ITERS=100 timeout -k 10 120 $B | grep "round 2\|max abs\|stamps\|FAIL" | cut -c1-190; }
for sh in 1 6 9; do run LT_SHAPE=$sh; done
run LT_SHAPE=9 LT_SPLITS=1
run LT_SHAPE=1 LT_SPLITS=2
run LT_SHAPE=16
run LT_SHAPE=16 LT_DBG=1
run LT_SHAPE=16 LT_CONST=1
for d in 2 4 6 8 14 16; do run LT_SHAPE=9 LT_DBG=$d; done
run LT_SHAPE=1 LT_CONST=1
for m in 2048 4096 8192 10240 10242; do run LT_SHAPE=1 LT_M=$m; done
