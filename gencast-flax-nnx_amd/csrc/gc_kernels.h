// Host-callable launchers of the gfx950 kernels (implemented in gc_kernels.hip).
// All pointers are device pointers; every launch goes to `stream` and returns
// immediately.  Row index convention: row = item * B + b  (item = node or edge).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GC_KERNELS_NS gc
#include "gc_kernels_decl.inc"
#undef GC_KERNELS_NS
#define GC_KERNELS_NS gc_a16
#include "gc_kernels_decl.inc"
#undef GC_KERNELS_NS

