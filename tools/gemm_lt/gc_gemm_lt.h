// gc_gemm_lt: the large-tile f16x3 GEMM of the 1-degree sizes -- BOTH operands shared through LDS.
// (Implemented in gc_gemm_lt.hip; the weight-streaming form it replaces there is gc_gemm_ws in gc_kernels.hip.)
//
// Operand images ("fragment order": what a lane feeds to v_mfma_f32_32x32x16_f16 is 16 contiguous bytes, a wave's
// operand 1 KB, so global -> LDS is a linear copy by global_load_lds_dwordx4 and LDS -> register a linear,
// conflict-free ds_read_b128):
//   WF16  weights, built once by gc_finalize (gc_api.hip encode_wf16): [32-column tile][k16 step][hi | lo][lane][8 halfs]
//   AF16  activations, written by their producers: [32-row tile][k16 step][hi | lo][lane][8 halfs]  (hi only when the
//         activation is exact fp16: "fp16 node features", the A16 kernel variants)
// Inside a k16 step lane (r, hk) (r = lane & 31 = row / column of the tile, hk = lane >> 5) holds the 8 values
//   k = 16 s + 8 hk + i             "natural" order   (h: written by the row passes; W_qkv, W_1)
//   k = 16 s + 8 (i >> 2) + 4 hk + (i & 3)   "permuted" order (the FFW hidden activation: what a lane of the FFW-1
//         epilogue holds after a transposed product is 8 such values = ONE 16-byte store; W_2 is encoded to match)
// A dot product does not care in which order k is visited as long as both operands agree.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gc_lt {

enum LtEpilogue : int {
  LT_EPI_F32 = 0,    // out[z][rows][ldo] float32 (+ bias, + gelu when act): slabs / per-node pre-activation terms
  LT_EPI_AF16 = 1,   // gelu(acc + bias) -> AF16 image in permuted k order (FFW layer 1 -> layer 2)
  LT_EPI_H16 = 2,    // A16 build only: gelu(acc + bias) -> _Float16 [rows][ldo] row-major (what the weight-streaming FFW-2 reads)
  LT_EPI_QKV = 3,    // q -> out (float32 [rows][ldo], halfs in the A16 build), k / v -> kv16 planes (as gc_gemm_ws epi 3)
};

struct LtArgs {
  const void* a;       // AF16 image of the activations, row tiles padded to a multiple of 8 (256 rows)
  int a_steps;         // k16 steps per row tile in the image (full K / 16)
  const float* wt;     // WF16 image of W^T
  int w_steps;         // k16 steps per column tile in the image (full K / 16)
  int rows, n;         // valid rows; n % 128 == 0
  int k_steps;         // k16 steps of one split (multiple of the kernel's steps per stage); split z starts at z * k_steps
  int splits;
  const float* bias;   // [n] or nullptr
  int act;             // 1: gelu(tanh)
  float* out;          // see LtEpilogue
  int ldo;
  int out_steps;       // LT_EPI_AF16: k16 steps per row tile of the output image (n / 16)
  int round16;         // fp16-feature mode: outputs are rounded to fp16 where the mode rounds them
  void* kv16;          // LT_EPI_QKV
  int kv_d;
  int out_f32;         // A16 build, LT_EPI_F32: always float32 (kept for symmetry with GemmArgs)
  int dbg;             // diagnostics (tools/bench_gemm_lt): bit 0 = every workgroup copies the operands of tile (0, 0)
  unsigned long long* stamps;   // diagnostics, shape 16: 4 words per workgroup (see gc_gemm_lt2_kernel)
  int shape;           // tile / pipeline variant (0: default; see launch_t in gc_gemm_lt.hip)
  int gx;              // XCD column groups (1, 2, 4 or 8; (n / 128) % gx == 0): see the kernel's tile map
};

// cls: a gc::KernelClass value (kernel-name suffix for the per-class profile).  a16: the activation image holds the
// hi plane only (exact fp16 values).  Returns hipErrorInvalidValue for shapes the kernel does not cover.
hipError_t launch_gemm_lt(hipStream_t s, int cls, const LtArgs& g, int epi, bool a16);
// bytes per row tile of an AF16 image with `steps` k16 steps
inline size_t af16_tile_bytes(int steps, bool a16) { return (size_t)steps * (a16 ? 1 : 2) * 1024; }
inline int lt_row_tiles(int rows) { return ((rows + 255) / 256) * 8; }   // allocation: padded to 256 rows

}  // namespace gc_lt
