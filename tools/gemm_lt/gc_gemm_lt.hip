// gfx950 (MI355X / CDNA4): the large-tile f16x3 GEMM -- both operands shared through LDS (gc_gemm_lt.h).
//
// Reference ops: the dense per-node projections of the mesh transformer, sparse_transformer.py:252-268 (FeedForward:
// Linear -> gelu -> Linear) and :271-290 (the Q, K, V projections), at the 1-degree sizes (10 242 mesh rows, d_model
// 512, hidden 2 048) where the weight-streaming kernel gc_gemm_ws (gc_kernels.hip) sat at 0.3 of the f16x3 MFMA
// ceiling: each of its waves pulled its own weight fragments L2 -> registers and used them for two row tiles only
// (~1 GB of L2 -> CU traffic per FFW-1 launch for 110 MB of operands), and split the float32 activation tile to
// hi / lo halfs with vector instructions the matrix pipe could not overlap.
//
// STATUS: an EXPERIMENT, not part of libgencast_hip.so (moved out of csrc/ in round 5).  Three kernel generations below,
// all parity-green in round 4 (inside the forward behind GC_TUNE_GEMM_LT=1, and in the harness next to this file): 4-18 %
// faster per launch than gc_gemm_ws, not faster end to end -- the in-kernel stamps and ablations of bench_gemm_lt.cpp say why
// (profiles/r04_lt_gemm_harness.txt).  The product side of the experiment (row passes writing AF16
// images, the forward's opt-in branches, the permuted W_2 image) was removed with it: git history of round 4 has it.
//
// Generation 1: a workgroup (4 waves, 2 x 2) owns a 128 x 128 output tile; per 32-deep K stage it copies 4 row-tile blocks of
// the activation image and 4 column-tile blocks of the weight image global -> LDS with global_load_lds_dwordx4 (1 KB
// per wave-instruction, linear on both sides: both images are in MFMA fragment order), two stages in LDS, ONE raw
// s_barrier per stage; every wave reads the fragments of its 64 x 64 sub-tile with conflict-free linear ds_read_b128
// (8 reads per 12 MFMAs) and runs the three-MFMA split product.  No vector arithmetic at all inside the K loop: the
// activations arrive already split (the row passes and the FFW-1 epilogue write the AF16 image).  Two workgroups per
// CU (64 KB of LDS, <= 256 registers) cover each other's barriers and epilogues.
#include "gc_gemm_lt.h"

#include <stdlib.h>
#include <atomic>
#include <type_traits>

#include "gc_kernels.h"

namespace gc_lt {

#include "gc_dev_common.inc"

// ---- AF16: activations in MFMA fragment order (gc_gemm_lt.h) -------------------------------------------------------
// [32-row tile][k16 step][hi | lo][lane = (k % 16 / 8) * 32 + row % 32][8 halfs], natural k order; `steps` = k16 steps
// per row tile.  A producer that owns four consecutive columns c .. c + 3 (c % 4 == 0) of a row writes one 8-byte
// piece per plane.  The exact-fp16 form (fp16 node features) has the hi plane only.
__device__ __forceinline__ void store4_af16(float* img, size_t row, int steps, int c, f32x4 v) {
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  _Float16 h[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) split16(v[e], h[e], l[e]);
  _Float16* p = reinterpret_cast<_Float16*>(img) + (((row >> 5) * steps + (c >> 4)) * 2) * 512 +
                ((((c >> 3) & 1) * 32 + (row & 31)) * 8) + (c & 4);
  *reinterpret_cast<f16x4*>(p) = f16x4{h[0], h[1], h[2], h[3]};
  *reinterpret_cast<f16x4*>(p + 512) = f16x4{l[0], l[1], l[2], l[3]};
}
__device__ __forceinline__ void store4_af16_hi(float* img, size_t row, int steps, int c, f32x4 v) {
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  _Float16* p = reinterpret_cast<_Float16*>(img) + ((row >> 5) * steps + (c >> 4)) * 512 +
                ((((c >> 3) & 1) * 32 + (row & 31)) * 8) + (c & 4);
  *reinterpret_cast<f16x4*>(p) = f16x4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
}


typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const char* gsrc, char* ldst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)ldst, 16, 0, 0);
}

// s_waitcnt vmcnt(n) with a run-time n (the tail of the K loop; the steady state uses an immediate)
__device__ __forceinline__ void wait_vm(int n) {
  switch (n) {
#define GC_VM(N_) case N_: asm volatile("s_waitcnt vmcnt(" #N_ ")" ::: "memory"); break;
    GC_VM(0) GC_VM(1) GC_VM(2) GC_VM(3) GC_VM(4) GC_VM(5) GC_VM(6) GC_VM(7) GC_VM(8) GC_VM(9) GC_VM(10) GC_VM(11) GC_VM(12)
    GC_VM(13) GC_VM(14) GC_VM(15) GC_VM(16) GC_VM(17) GC_VM(18) GC_VM(19) GC_VM(20) GC_VM(21) GC_VM(22) GC_VM(23) GC_VM(24)
    GC_VM(25) GC_VM(26) GC_VM(27) GC_VM(28) GC_VM(29) GC_VM(30) GC_VM(31) GC_VM(32) GC_VM(33) GC_VM(34) GC_VM(35) GC_VM(36)
    GC_VM(37) GC_VM(38) GC_VM(39) GC_VM(40) GC_VM(41) GC_VM(42) GC_VM(43) GC_VM(44) GC_VM(45) GC_VM(46) GC_VM(47) GC_VM(48)
#undef GC_VM
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}
template <int N>
__device__ __forceinline__ void wait_vm_c() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Epilogues, shared by the kernel generations below.  acc / acc2: the wave's 2 x 2 accumulator tiles (hi x hi, and the
// two cross terms scaled by 2048); row0 / col0: first row / column of the wave's 64 x 64 sub-tile; z: K split.
template <int EPI, bool A16>
__device__ __forceinline__ void lt_epilogue(const LtArgs& g, f32x16 (&acc)[2][2], f32x16 (&acc2)[2][2], int row0, int col0,
                                            int z, int lane) {
  const int r = lane & 31, hh = lane >> 5;
  if constexpr (EPI == LT_EPI_F32) {
    // lane (r, hh): column col0 + 32 j + r, rows row0 + 32 i + acc_row(q, hh)
    float* obase = g.out + (size_t)z * g.rows * g.ldo + col0 + r;
    float bias_v[2] = {0.f, 0.f};
    if (g.bias) {
      bias_v[0] = g.bias[col0 + r];
      bias_v[1] = g.bias[col0 + 32 + r];
    }
    const bool full = row0 + 64 <= g.rows;
    auto emit = [&](auto full_c, auto act_c) __attribute__((always_inline)) {
      constexpr bool FULL = decltype(full_c)::value, ACT = decltype(act_c)::value;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            float v = acc[i][j][q] + acc2[i][j][q] * (1.0f / kLoScale) + bias_v[j];
            if (ACT) v = gelu_tanh_fast(v);
            const int grow = row0 + 32 * i + acc_row(q, hh);
            if (FULL || grow < g.rows) obase[(size_t)grow * g.ldo + 32 * j] = v;
          }
    };
    if (full && g.act) emit(std::true_type{}, std::true_type{});
    else if (full) emit(std::true_type{}, std::false_type{});
    else if (g.act) emit(std::false_type{}, std::true_type{});
    else emit(std::false_type{}, std::false_type{});
    return;
  }
  if constexpr (EPI == LT_EPI_AF16) {
    // transposed product: lane (r, hh) holds row row0 + 32 i + r, columns col0 + 32 j + 8 jj + 4 hh + e in register
    // 4 jj + e.  Registers of jj = 2 s2, 2 s2 + 1 are the lane's 8 values of k16 step (col0 + 32 j) / 16 + s2 of
    // the output image in PERMUTED k order: one 16-byte store per plane, 1 KB contiguous per wave-instruction.
    constexpr int PO = A16 ? 1 : 2;
    char* obase = reinterpret_cast<char*>(g.out) + lane * 16;
    with_flag(g.act, [&](auto act_c) __attribute__((always_inline)) {
      constexpr bool ACT = decltype(act_c)::value;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int cb = col0 + 32 * j + 4 * hh;
        f32x4 bv[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) bv[jj] = g.bias ? ld4(g.bias + cb + 8 * jj) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const size_t rtile = (size_t)((row0 >> 5) + i);
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            _Float16 hi[8], lo[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
              const int jj = 2 * s2 + (t >> 2), e = t & 3;
              float x = acc[i][j][4 * jj + e] + acc2[i][j][4 * jj + e] * (1.0f / kLoScale) + bv[jj][e];
              if (ACT) x = gelu_tanh_fast(x);
              if constexpr (A16) hi[t] = (_Float16)x;
              else split16(x, hi[t], lo[t]);
            }
            char* dst = obase + ((rtile * g.out_steps + (size_t)((col0 + 32 * j) >> 4) + s2) * PO) * 1024;
            if ((g.dbg & 8) && hi[0] != (_Float16)12345.f) continue;   // (diagnostic bit 3: the arithmetic without the stores)
            const f16x8 vh = f16x8{hi[0], hi[1], hi[2], hi[3], hi[4], hi[5], hi[6], hi[7]};
            const f16x8 vl = f16x8{lo[0], lo[1], lo[2], lo[3], lo[4], lo[5], lo[6], lo[7]};
            if (g.dbg & 16) {                    // (diagnostic bit 4: write-through stores, dropped from this XCD's L2)
              asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(dst), "v"(vh) : "memory");
              if constexpr (!A16) asm volatile("global_store_dwordx4 %0, %1, off offset:1024 sc1" : : "v"(dst), "v"(vl) : "memory");
              asm volatile("s_nop 1");
              continue;
            }
            if (g.dbg & 32) {                    // (diagnostic bit 5: non-temporal stores)
              __builtin_nontemporal_store(vh, reinterpret_cast<f16x8*>(dst));
              if constexpr (!A16) __builtin_nontemporal_store(vl, reinterpret_cast<f16x8*>(dst + 1024));
              continue;
            }
            *reinterpret_cast<f16x8*>(dst) = vh;
            if constexpr (!A16) *reinterpret_cast<f16x8*>(dst + 1024) = vl;
          }
        }
      }
    });
    return;
  }
  if constexpr (EPI == LT_EPI_H16) {
    // fp16-stored row-major output (gc_gemm_ws's A16 epilogue 0): lane (r, hh) holds columns col0 + 32 j + 8 jj + 4 hh + e
    // of row row0 + 32 i + r: bias (+ gelu), one 8-byte store per group of four
    static_assert(A16, "half storage");
    _Float16* ob = as_h16(g.out);
    with_flag(g.act, [&](auto act_c) __attribute__((always_inline)) {
      constexpr bool ACT = decltype(act_c)::value;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int cb = col0 + 32 * j + 4 * hh;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const f32x4 bv = g.bias ? ld4(g.bias + cb + 8 * jj) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int grow = row0 + 32 * i + r;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float x = acc[i][j][4 * jj + e] + acc2[i][j][4 * jj + e] * (1.0f / kLoScale) + bv[e];
              if (ACT) x = gelu_tanh_fast(x);
              v[e] = x;
            }
            if (grow < g.rows) sth4(ob + (size_t)grow * g.ldo + cb + 8 * jj, v);
          }
        }
      }
    });
    return;
  }
  if constexpr (EPI == LT_EPI_QKV) {
    // as gc_gemm_ws epilogue 3: q float32 (halfs in the A16 build), k and v as fp16 hi / lo planes of kv16
    const int D = g.kv_d;
    _Float16* kv = reinterpret_cast<_Float16*>(g.kv16);
    with_flag(g.round16, [&](auto rc) __attribute__((always_inline)) {
      constexpr bool RND = decltype(rc)::value;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int gcol = col0 + 32 * j;           // first column of this 32-column tile in [0, 3 D)
        const int which = gcol / D, col_in = gcol - which * D;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int grow = row0 + 32 * i + r;
          if (grow >= g.rows) continue;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float x = acc[i][j][4 * jj + e] + acc2[i][j][4 * jj + e] * (1.0f / kLoScale);
              x = (fabsf(x) <= kF16Max) ? x : __builtin_nanf("");     // leaves the f16x3 domain here or never
              v[e] = r16_c<RND>(x);
            }
            const int c = col_in + 8 * jj + 4 * hh;
            if (which == 0) {
              if constexpr (A16) sth4(as_h16(g.out) + (size_t)grow * g.ldo + c, v);
              else st4(g.out + (size_t)grow * g.ldo + c, v);
            } else if constexpr (A16) {
              sth4(kv + (size_t)grow * (4 * D) + (size_t)(which - 1) * (2 * D) + c, v);
            } else {
              _Float16 h4[4], l4[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) split16(v[e], h4[e], l4[e]);
              _Float16* dst = kv + (size_t)grow * (4 * D) + (size_t)(which - 1) * (2 * D) + c;
              *reinterpret_cast<f16x4*>(dst) = f16x4{h4[0], h4[1], h4[2], h4[3]};
              *reinterpret_cast<f16x4*>(dst + D) = f16x4{l4[0], l4[1], l4[2], l4[3]};
            }
          }
        }
      }
    });
    return;
  }
}

// ---- the consumer's stage body as inline asm (used by both kernel generations; why: see generation 2's header) ----
#define LT_RD(dst, base, off) "ds_read_b128 %[" #dst "], %[" #base "] offset:" #off "\n\t"
// one 32 x 32 x 16 product step: D += A x B (operand order swapped for the transposed product)
#define LT_MF_N(d, a, b) "v_mfma_f32_32x32x16_f16 %[" #d "], %[" #a "], %[" #b "], %[" #d "]\n\t"
#define LT_MF_T(d, a, b) "v_mfma_f32_32x32x16_f16 %[" #d "], %[" #b "], %[" #a "], %[" #d "]\n\t"
// the 12 MFMAs of one k16 step on fragment set P (x / y): c = hi x hi, d = hi x lo + lo x hi; dependent pairs kept apart
#define LT_MMA3(MF, P)                                                                                     \
  MF(d00, P##ah0, P##bl0) MF(c00, P##ah0, P##bh0) MF(d01, P##ah0, P##bl1) MF(c01, P##ah0, P##bh1)            \
  MF(d10, P##ah1, P##bl0) MF(c10, P##ah1, P##bh0) MF(d11, P##ah1, P##bl1) MF(c11, P##ah1, P##bh1)            \
  MF(d00, P##al0, P##bh0) MF(d01, P##al0, P##bh1) MF(d10, P##al1, P##bh0) MF(d11, P##al1, P##bh1)
// ... and the 8 of the exact-fp16 activation form (no lo plane)
#define LT_MMA2(MF, P)                                                                                     \
  MF(d00, P##ah0, P##bl0) MF(c00, P##ah0, P##bh0) MF(d01, P##ah0, P##bl1) MF(c01, P##ah0, P##bh1)            \
  MF(d10, P##ah1, P##bl0) MF(c10, P##ah1, P##bh0) MF(d11, P##ah1, P##bl1) MF(c11, P##ah1, P##bh1)
// fragment reads of one k16 step into set P: float32-feature image (A: hi, lo per step = 2 KB; row tile stride 4 KB)
#define LT_READS3(P, o0, o1, o2, o3)                                                                       \
  LT_RD(P##ah0, a, o0) LT_RD(P##bh0, b, o0) LT_RD(P##bl0, b, o1) LT_RD(P##al0, a, o1)                         \
  LT_RD(P##ah1, a, o2) LT_RD(P##bh1, b, o2) LT_RD(P##bl1, b, o3) LT_RD(P##al1, a, o3)
// exact-fp16 activations: A has the hi plane only (1 KB per step, row tile stride 2 KB); B as above
#define LT_READS2(P, a0, a1, o0, o1, o2, o3)                                                               \
  LT_RD(P##ah0, a, a0) LT_RD(P##bh0, b, o0) LT_RD(P##bl0, b, o1)                                             \
  LT_RD(P##ah1, a, a1) LT_RD(P##bh1, b, o2) LT_RD(P##bl1, b, o3)

struct LtFrag { f32x4 ah0, ah1, al0, al1, bh0, bh1, bl0, bl1; };

#define LT_ASM_OPERANDS                                                                                              \
  : [c00] "+v"(acc[0][0]), [c01] "+v"(acc[0][1]), [c10] "+v"(acc[1][0]), [c11] "+v"(acc[1][1]),                       \
    [d00] "+v"(acc2[0][0]), [d01] "+v"(acc2[0][1]), [d10] "+v"(acc2[1][0]), [d11] "+v"(acc2[1][1]),                   \
    [yah0] "+v"(Y.ah0), [yah1] "+v"(Y.ah1), [yal0] "+v"(Y.al0), [yal1] "+v"(Y.al1), [ybh0] "+v"(Y.bh0),               \
    [ybh1] "+v"(Y.bh1), [ybl0] "+v"(Y.bl0), [ybl1] "+v"(Y.bl1), [xah0] "=&v"(X.ah0), [xah1] "=&v"(X.ah1),             \
    [xal0] "=&v"(X.al0), [xal1] "=&v"(X.al1), [xbh0] "=&v"(X.bh0), [xbh1] "=&v"(X.bh1), [xbl0] "=&v"(X.bl0),          \
    [xbl1] "=&v"(X.bl1)                                                                                              \
  : [a] "v"(a_addr), [b] "v"(b_addr)                                                                                 \
  : "memory"

// One stage of the consumer (SPB = 2 k16 steps; set Y holds the previous step's fragments on entry, this stage's
// second step's on exit): reads X <- step 0 | MFMAs on Y | reads Y <- step 1 | wait for X | MFMAs on X.
// FIRST: no previous step (Y is not multiplied).  LAST form is lt_consumer_drain.
template <bool A16, bool TR, bool FIRST>
__device__ __forceinline__ void lt_consumer_stage(f32x16 (&acc)[2][2], f32x16 (&acc2)[2][2], LtFrag& Y, unsigned a_addr,
                                                  unsigned b_addr) {
  LtFrag X;
#define LT_STAGE(READS_X, MMA_Y, READS_Y, WAITX, MMA_X) asm volatile(READS_X MMA_Y READS_Y WAITX MMA_X LT_ASM_OPERANDS)
  if constexpr (!A16) {
    if constexpr (FIRST && TR) {
      LT_STAGE(LT_READS3(x, 0, 1024, 4096, 5120), "", LT_READS3(y, 2048, 3072, 6144, 7168), "s_waitcnt lgkmcnt(8)\n\t",
               LT_MMA3(LT_MF_T, x));
    } else if constexpr (FIRST) {
      LT_STAGE(LT_READS3(x, 0, 1024, 4096, 5120), "", LT_READS3(y, 2048, 3072, 6144, 7168), "s_waitcnt lgkmcnt(8)\n\t",
               LT_MMA3(LT_MF_N, x));
    } else if constexpr (TR) {
      LT_STAGE(LT_READS3(x, 0, 1024, 4096, 5120), LT_MMA3(LT_MF_T, y), LT_READS3(y, 2048, 3072, 6144, 7168),
               "s_waitcnt lgkmcnt(8)\n\t", LT_MMA3(LT_MF_T, x));
    } else {
      LT_STAGE(LT_READS3(x, 0, 1024, 4096, 5120), LT_MMA3(LT_MF_N, y), LT_READS3(y, 2048, 3072, 6144, 7168),
               "s_waitcnt lgkmcnt(8)\n\t", LT_MMA3(LT_MF_N, x));
    }
  } else {
    if constexpr (FIRST && TR) {
      LT_STAGE(LT_READS2(x, 0, 2048, 0, 1024, 4096, 5120), "", LT_READS2(y, 1024, 3072, 2048, 3072, 6144, 7168),
               "s_waitcnt lgkmcnt(6)\n\t", LT_MMA2(LT_MF_T, x));
    } else if constexpr (FIRST) {
      LT_STAGE(LT_READS2(x, 0, 2048, 0, 1024, 4096, 5120), "", LT_READS2(y, 1024, 3072, 2048, 3072, 6144, 7168),
               "s_waitcnt lgkmcnt(6)\n\t", LT_MMA2(LT_MF_N, x));
    } else if constexpr (TR) {
      LT_STAGE(LT_READS2(x, 0, 2048, 0, 1024, 4096, 5120), LT_MMA2(LT_MF_T, y),
               LT_READS2(y, 1024, 3072, 2048, 3072, 6144, 7168), "s_waitcnt lgkmcnt(6)\n\t", LT_MMA2(LT_MF_T, x));
    } else {
      LT_STAGE(LT_READS2(x, 0, 2048, 0, 1024, 4096, 5120), LT_MMA2(LT_MF_N, y),
               LT_READS2(y, 1024, 3072, 2048, 3072, 6144, 7168), "s_waitcnt lgkmcnt(6)\n\t", LT_MMA2(LT_MF_N, x));
    }
  }
#undef LT_STAGE
}
// the last step's MFMAs (fragments in Y, read by the last stage), then the wait states an 8-pass MFMA result needs
// before anything but an accumulating MFMA touches it (cdna_hip_programming.md section 5.7 item 2)
template <bool A16, bool TR>
__device__ __forceinline__ void lt_consumer_drain(f32x16 (&acc)[2][2], f32x16 (&acc2)[2][2], LtFrag& Y) {
  LtFrag X;
  const unsigned a_addr = 0, b_addr = 0;
  if constexpr (!A16) {
    if constexpr (TR) asm volatile("s_waitcnt lgkmcnt(0)\n\t" LT_MMA3(LT_MF_T, y) "s_nop 15\n\t" LT_ASM_OPERANDS);
    else asm volatile("s_waitcnt lgkmcnt(0)\n\t" LT_MMA3(LT_MF_N, y) "s_nop 15\n\t" LT_ASM_OPERANDS);
  } else {
    if constexpr (TR) asm volatile("s_waitcnt lgkmcnt(0)\n\t" LT_MMA2(LT_MF_T, y) "s_nop 15\n\t" LT_ASM_OPERANDS);
    else asm volatile("s_waitcnt lgkmcnt(0)\n\t" LT_MMA2(LT_MF_N, y) "s_nop 15\n\t" LT_ASM_OPERANDS);
  }
}

// WM: waves along M (2: 128-row tile, 256 threads; 4: 256-row tile, 512 threads); always 2 waves along N (128 columns),
// every wave a 64 x 64 sub-tile.  SPB: k16 steps per stage; NST: stages in the LDS ring (NST - 1 in flight ahead of the
// one being read).  LDS per stage: A 2 WM row tiles x SPB x (hi, lo) KB, B 4 column tiles x SPB x 2 KB.
template <int EPI, int CLS, bool A16, int WM, int SPB, int NST, int OCC>
__global__ __launch_bounds__(128 * WM, OCC) void gc_gemm_lt_kernel(LtArgs g) {
  constexpr int PA = A16 ? 1 : 2;
  constexpr int NW = 2 * WM;                    // waves
  constexpr int BM = 64 * WM;
  constexpr int A_WAVE = SPB * PA * 1024;       // bytes one wave copies per stage for A (its row tile)
  constexpr int B_TILE = SPB * 2 * 1024;        // bytes of one column tile per stage
  constexpr int B_WAVE = 4 * B_TILE / NW;       // bytes one wave copies per stage for B
  constexpr int A_STAGE = NW * A_WAVE, STAGE = A_STAGE + 4 * B_TILE;
  constexpr int GPW = (A_WAVE + B_WAVE) / 1024; // copy instructions per wave per stage
  constexpr int PF = NST - 1;                   // stages in flight ahead
  constexpr bool TR = EPI != LT_EPI_F32;   // (AF16, H16, QKV epilogues)        // transposed product: a lane owns 4 consecutive columns of a row
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- tile of this workgroup.  Workgroups are dealt to the 8 XCDs round-robin (blockIdx.x % 8).  The XCDs form
  // gx column groups x (8 / gx) row groups: an XCD works on n / 128 / gx column tiles -- their weight blocks stay
  // in its 4-MB L2 for the whole launch -- and on a contiguous range of row tiles, row-tile-major, so that the
  // column tiles of one row tile run together and the activation blocks come from memory once per XCD.
  const int n_rt = (g.rows + BM - 1) / BM, n_ct = g.n >> 7;
  int rt, ct, z;
  {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int gxi = xcd % g.gx, gyi = xcd / g.gx, GY = 8 / g.gx;
    const int n_rz = n_rt * g.splits;
    const int base = n_rz / GY, extra = n_rz - base * GY;
    const int cnt = base + (gyi < extra ? 1 : 0);
    const int first = gyi * base + (gyi < extra ? gyi : extra);
    const int cpg = n_ct / g.gx;
    if (j >= cnt * cpg) return;
    const int jr = j / cpg;
    const int rz = first + jr;
    ct = gxi * cpg + (j - jr * cpg);
    rt = rz / g.splits;
    z = rz - rt * g.splits;
  }
  const int k0 = z * g.k_steps;
  const int nst = g.k_steps / SPB;

  // ---- copy roles: wave w moves row tile NW rt + w of A (a stage's SPB steps are contiguous in the image) and its
  // share of the 4 column tiles of B: column tile w & 3, bytes [(w >> 2) B_WAVE, +B_WAVE) of the tile's stage
  const char* a_src = reinterpret_cast<const char*>(g.a) +
                      ((size_t)(rt * NW + wave) * g.a_steps + k0) * (PA * 1024) + lane * 16;
  const char* b_src = reinterpret_cast<const char*>(g.wt) +
                      ((size_t)(ct * 4 + (wave & 3)) * g.w_steps + k0) * 2048 + (wave >> 2) * B_WAVE + lane * 16;
  const int a_dst = wave * A_WAVE;
  const int b_dst = A_STAGE + (wave & 3) * B_TILE + (wave >> 2) * B_WAVE;
  auto issue = [&](int st, int buf) __attribute__((always_inline)) {
    const char* as = a_src + (size_t)st * A_WAVE;
    const char* bs = b_src + (size_t)st * B_TILE;
    char* base = lds + buf * STAGE;
#pragma unroll
    for (int p = 0; p < A_WAVE / 1024; ++p) glds16(as + p * 1024, base + a_dst + p * 1024);
#pragma unroll
    for (int p = 0; p < B_WAVE / 1024; ++p) glds16(bs + p * 1024, base + b_dst + p * 1024);
  };

  f32x16 acc[2][2], acc2[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        acc[i][j][q] = 0.f;
        acc2[i][j][q] = 0.f;
      }

  // fragments of this wave's 64 x 64 sub-tile: row tiles 2 wm + i, column tiles 2 wn + j
  const int a_frag = (wm * 2) * A_WAVE + lane * 16;
  const int b_frag = A_STAGE + (wn * 2) * B_TILE + lane * 16;
  // ---- K loop over a ring of NST stage buffers: stage `it` is read from buffer it % NST while stages it + 1 ..
  // it + PF land in the others.  The counted wait + raw barrier at the top of an iteration say "every wave's copies of
  // stage `it` have landed AND every wave is done reading stage it - 1", which frees the buffer stage it + PF goes
  // to.  The copies stay in flight across the barrier (s_waitcnt vmcnt(N) with N = the copies issued for later
  // stages, never __syncthreads()).
  // The fragment reads run ONE k16 step ahead of the MFMAs, across the barrier: after the barrier a wave first reads
  // the fragments of the new stage's first step, then issues the MFMAs of the PREVIOUS stage's last step (operands in
  // registers since before the barrier), so no MFMA group waits for an LDS round trip.  (First version: reads, wait,
  // MFMAs per step -- four exposed LDS latencies per stage; the matrix pipe was 53 % busy while waves were resident.)
  static_assert(SPB == 2, "lt_consumer_stage is written for two k16 steps per stage");
#pragma unroll
  for (int st = 0; st < PF; ++st)
    if (st < nst) issue(st, st);
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  LtFrag Y{z4, z4, z4, z4, z4, z4, z4, z4};
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const unsigned a_addr0 = lds0 + a_frag, b_addr0 = lds0 + b_frag;
  int buf = 0, ibuf = PF % NST;                 // buffer read by stage `it` / written for stage it + PF
  for (int it = 0; it < nst; ++it) {
    const int later = nst - 1 - it;             // stages after `it` whose copies are in flight: min(PF - 1, later)
    if (later >= PF - 1) wait_vm_c<(PF - 1) * GPW>();
    else wait_vm(later * GPW);
    // this wave's reads of stage it - 1 are done (its second step's were issued 12 MFMAs ago); the operands pin the
    // fragment registers across the statement boundary
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(Y.ah0), "+v"(Y.ah1), "+v"(Y.al0), "+v"(Y.al1), "+v"(Y.bh0), "+v"(Y.bh1), "+v"(Y.bl0), "+v"(Y.bl1)
                 :
                 : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (it + PF < nst) issue(it + PF, ibuf);
    ibuf = ibuf + 1 == NST ? 0 : ibuf + 1;
    if (it == 0) lt_consumer_stage<A16, TR, true>(acc, acc2, Y, a_addr0, b_addr0);
    else lt_consumer_stage<A16, TR, false>(acc, acc2, Y, a_addr0 + buf * STAGE, b_addr0 + buf * STAGE);
    buf = buf + 1 == NST ? 0 : buf + 1;
  }
  lt_consumer_drain<A16, TR>(acc, acc2, Y);

  // ---- epilogues
  lt_epilogue<EPI, A16>(g, acc, acc2, rt * BM + wm * 64, ct * 128 + wn * 64, z, lane);
}

// ----------------------------------------------------------------------------------------------------------------
// Generation 2: producer / consumer waves.  What the counters said about generation 1 (profiles/r04_lt_*): the matrix
// pipe was 41 % busy at full clock although copies, fragment reads and MFMAs all fit -- a wave that issues a 1-KB
// global_load_lds waits 100-200 cycles for the vector-memory path to take it, in order, with its MFMAs queued behind
// (SQ_WAIT_INST_ANY = 53 % of wave cycles); deeper rings, larger tiles and other XCD maps all landed on the same time.
// Here a workgroup is 8 waves on one CU: waves 0-3 (one per SIMD) only read fragments and issue MFMAs, waves 4-7 (their
// SIMD partners) only issue the copies, NST - 1 stages ahead, and wait for them; ONE s_barrier per stage joins the two
// roles ("stage it has landed" from the producers, "stage it - 1 is read" from the consumers).  The consumer's stage
// body is one asm statement (fragment reads one k16 step ahead of the MFMAs, counted lgkmcnt): hipcc sank the reads
// below the MFMAs that should cover them and waited lgkmcnt(0) on the spot (sched_group_barrier did not hold it).
// ----------------------------------------------------------------------------------------------------------------
// STAMP (diagnostic instantiation, LtArgs.shape 16 + LtArgs.stamps): consumer wave 0 of every workgroup records
// {shader cycles of the K loop, of which waiting at the stage barriers, of which inside the stage bodies, 100-MHz
// ticks of the K loop}; every stamp drains lgkmcnt (s_memtime is a scalar-memory read), so the loop runs a few percent
// slower than the product instantiation
#define LT_NOW(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory")
template <int EPI, int CLS, bool A16, int NST, bool STAMP = false>
__global__ __launch_bounds__(512, 2) void gc_gemm_lt2_kernel(LtArgs g) {
  constexpr int PA = A16 ? 1 : 2, SPB = 2;
  constexpr int A_WAVE = SPB * PA * 1024;       // bytes of one row tile per stage
  constexpr int B_TILE = SPB * 2 * 1024;        // bytes of one column tile per stage
  constexpr int A_STAGE = 4 * A_WAVE, STAGE = A_STAGE + 4 * B_TILE;
  constexpr int GPW = (A_WAVE + B_TILE) / 1024; // copy instructions per producer wave per stage
  constexpr int PF = NST - 1;
  constexpr bool TR = EPI != LT_EPI_F32;   // (AF16, H16, QKV epilogues)
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  // ---- PERSISTENT: the grid is one workgroup per CU (8 x wpx workgroups; blockIdx.x % 8 = XCD under round-robin
  // placement -- speed only, any placement computes the same tiles).  XCD x owns the tile list of the generation-1
  // map (gx column groups x 8 / gx row groups, row-tile-major); workgroup w of the XCD takes tiles w, w + wpx, ...
  // Both roles walk the same list; the ring of stage buffers does not drain between tiles, so the producers are
  // NST - 1 stages into the next tile while the consumers run the epilogue of this one.
  const int n_rt = (g.rows + 127) >> 7, n_ct = g.n >> 7;
  const int xcd = blockIdx.x & 7, wslot = blockIdx.x >> 3, wpx = gridDim.x >> 3;
  const int gxi = xcd % g.gx, gyi = xcd / g.gx, GY = 8 / g.gx;
  const int n_rz = n_rt * g.splits;
  const int rz_base = n_rz / GY, rz_extra = n_rz - rz_base * GY;
  const int rz_cnt = rz_base + (gyi < rz_extra ? 1 : 0);
  const int rz_first = gyi * rz_base + (gyi < rz_extra ? gyi : rz_extra);
  const int cpg = n_ct / g.gx;
  const int n_list = rz_cnt * cpg;              // tiles of this XCD
  const int n_mine = wslot < n_list ? (n_list - wslot + wpx - 1) / wpx : 0;
  if (n_mine == 0) return;
  const int nst = g.k_steps / SPB;
  auto tile_of = [&](int k, int& rt, int& ct, int& z) __attribute__((always_inline)) {
    const int j = wslot + k * wpx;
    const int jr = j / cpg;
    const int rz = rz_first + jr;
    ct = gxi * cpg + (j - jr * cpg);
    rt = rz / g.splits;
    z = rz - rt * g.splits;
  };

  if (wave >= 4) {
    // ---- producer p: row tile 4 rt + p of A and column tile 4 ct + p of B, every stage of every tile
    const int p = wave - 4;
    const int a_dst = p * A_WAVE, b_dst = A_STAGE + p * B_TILE;
    const int total = n_mine * nst;             // stages of this workgroup
    int ik = 0, ist = 0, ibuf = 0;              // issue cursor: tile, stage in the tile, ring buffer
    const char* a_src = nullptr;
    const char* b_src = nullptr;
    auto set_tile = [&](int k) __attribute__((always_inline)) {
      int rt, ct, z;
      tile_of(k, rt, ct, z);
      if (g.dbg & 1) { rt = 0; ct = 0; }        // diagnostic: every workgroup copies tile (0, 0)
      const int k0 = z * g.k_steps;
      a_src = reinterpret_cast<const char*>(g.a) + ((size_t)(rt * 4 + p) * g.a_steps + k0) * (PA * 1024) + lane * 16;
      b_src = reinterpret_cast<const char*>(g.wt) + ((size_t)(ct * 4 + p) * g.w_steps + k0) * 2048 + lane * 16;
    };
    auto issue_next = [&]() __attribute__((always_inline)) {
      char* base = lds + ibuf * STAGE;
#pragma unroll
      for (int q = 0; q < A_WAVE / 1024; ++q) glds16(a_src + q * 1024, base + a_dst + q * 1024);
#pragma unroll
      for (int q = 0; q < B_TILE / 1024; ++q) glds16(b_src + q * 1024, base + b_dst + q * 1024);
      a_src += A_WAVE;
      b_src += B_TILE;
      ibuf = ibuf + 1 == NST ? 0 : ibuf + 1;
      if (++ist == nst) {
        ist = 0;
        if (++ik < n_mine) set_tile(ik);
      }
    };
    set_tile(0);
    int issued = 0;
    for (; issued < PF && issued < total; ++issued) issue_next();
    for (int s = 0; s < total; ++s) {           // global stage s has landed when at most the later issued stages are in flight
      const int later = issued - 1 - s;
      if (later >= PF - 1) wait_vm_c<(PF - 1) * GPW>();
      else wait_vm(later * GPW);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (issued < total) {                     // into the buffer of stage s - 1: every consumer has passed the barrier
        issue_next();
        ++issued;
      }
    }
    return;
  }

  // ---- consumer (wm, wn): the 64 x 64 sub-tile of row tiles 2 wm + i, column tiles 2 wn + j
  const int wm = wave >> 1, wn = wave & 1;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const unsigned a_frag = lds0 + (wm * 2) * A_WAVE + lane * 16;
  const unsigned b_frag = lds0 + A_STAGE + (wn * 2) * B_TILE + lane * 16;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  LtFrag Y{z4, z4, z4, z4, z4, z4, z4, z4};
  unsigned long long t_begin = 0, t_a = 0, t_b = 0, t_c = 0, t_bar = 0, t_body = 0, t_epi = 0, rt_begin = 0;
  if constexpr (STAMP) {
    LT_NOW(t_begin);
    rt_begin = __builtin_amdgcn_s_memrealtime();
  }
  int buf = 0;
  for (int k = 0; k < n_mine; ++k) {
    int rt, ct, z;
    tile_of(k, rt, ct, z);
    f32x16 acc[2][2], acc2[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          acc[i][j][q] = 0.f;
          acc2[i][j][q] = 0.f;
        }
    if constexpr (STAMP) LT_NOW(t_a);
    __builtin_amdgcn_s_barrier();               // the tile's stage 0 has landed
    asm volatile("" ::: "memory");
    if constexpr (STAMP) { LT_NOW(t_b); t_bar += t_b - t_a; }
    lt_consumer_stage<A16, TR, true>(acc, acc2, Y, a_frag + buf * STAGE, b_frag + buf * STAGE);
    if constexpr (STAMP) { LT_NOW(t_c); t_body += t_c - t_b; }
    buf = buf + 1 == NST ? 0 : buf + 1;
    for (int it = 1; it < nst; ++it) {
      if constexpr (STAMP) LT_NOW(t_a);
      // this wave's reads of the previous stage are done (its second step's were issued 12 MFMAs ago); the operands
      // pin the fragment registers across the statement boundary
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(Y.ah0), "+v"(Y.ah1), "+v"(Y.al0), "+v"(Y.al1), "+v"(Y.bh0), "+v"(Y.bh1), "+v"(Y.bl0), "+v"(Y.bl1)
                   :
                   : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if constexpr (STAMP) { LT_NOW(t_b); t_bar += t_b - t_a; }
      lt_consumer_stage<A16, TR, false>(acc, acc2, Y, a_frag + buf * STAGE, b_frag + buf * STAGE);
      if constexpr (STAMP) { LT_NOW(t_c); t_body += t_c - t_b; }
      buf = buf + 1 == NST ? 0 : buf + 1;
    }
    lt_consumer_drain<A16, TR>(acc, acc2, Y);   // (waits for the last fragment reads: the ring may now be overwritten)
    if constexpr (STAMP) LT_NOW(t_a);
    lt_epilogue<EPI, A16>(g, acc, acc2, rt * 128 + wm * 64, ct * 128 + wn * 64, z, lane);
    if constexpr (STAMP) { LT_NOW(t_c); t_epi += t_c - t_a; }
  }
  if constexpr (STAMP) {
    LT_NOW(t_c);
    const unsigned long long rt_end = __builtin_amdgcn_s_memrealtime();
    if (g.stamps && wave == 0 && lane == 0) {
      unsigned long long* o = g.stamps + (size_t)blockIdx.x * 8;
      o[0] = t_c - t_begin; o[1] = t_bar; o[2] = t_body; o[3] = rt_end - rt_begin; o[4] = t_epi; o[5] = n_mine;
    }
  }
}

// ----------------------------------------------------------------------------------------------------------------
// Generation 3: 256 x 128 tiles, 8 consumer waves (two per SIMD) + 4 producer waves, persistent.
// The stamps of generation 2 (profiles/r04_lt_stamps.txt): its consumer K loop is within 10 % of the MFMA issue time,
// but a 128 x 128 tile needs 32 KB of operands per 32-deep stage and a CU takes in ~23 B per clock from L2 (the
// guide's 66-73 GB/s per CU), so the copies (22 k cycles per tile) outlast the MFMAs (12.3 k) plus the epilogue
// (5.2 k, one wave per SIMD).  A 256 x 128 tile moves 0.75 x the bytes per MFMA; its 8 consumer waves share a SIMD in
// pairs -- one wave's fragment reads and epilogue arithmetic run under the other's MFMAs -- which is also what lets a
// consumer live with ONE fragment register set (12 waves per CU leave 168 registers per lane: 128 accumulators + 32
// fragment registers): a fragment register is reloaded for the next k16 step right after the last MFMA that reads it.
// The stage barrier sits in the middle of a stage's second step, between the last use of the first registers and
// their reload from the next stage's buffer.
// ----------------------------------------------------------------------------------------------------------------
#define LT_WAIT(n) "s_waitcnt lgkmcnt(" #n ")\n\t"
// one k16 step, float32-feature form: 12 MFMAs in the order that frees registers early; R1..R4 = the reload groups
// (after MFMA 6: ah0 al0; after 8: bl0 bl1; after 10: bh0; after 12: bh1 ah1 al1), MID = what happens after MFMA 6
// before its reloads (nothing, or the stage barrier).  Counted waits: LDS returns in order, so lgkmcnt(n) = "all but
// the n youngest reads are done".
#define LT3_STEP3(MF, MID, R1, R2, R3, R4)                                                                     \
  LT_WAIT(5) MF(d00, ah0, bl0) LT_WAIT(3) MF(c00, ah0, bh0) MF(d01, ah0, bl1) LT_WAIT(2) MF(c01, ah0, bh1)       \
  MF(d00, al0, bh0) MF(d01, al0, bh1) MID R1 LT_WAIT(3) MF(d10, ah1, bl0) MF(d11, ah1, bl1) R2                  \
  MF(c10, ah1, bh0) LT_WAIT(4) MF(d10, al1, bh0) R3 MF(c11, ah1, bh1) MF(d11, al1, bh1) R4
// the same step with nothing reloaded (the last step of a tile): every read is older, plain waits suffice
#define LT3_STEP3_LAST(MF)                                                                                    \
  LT_WAIT(0) MF(d00, ah0, bl0) MF(c00, ah0, bh0) MF(d01, ah0, bl1) MF(c01, ah0, bh1) MF(d00, al0, bh0)          \
  MF(d01, al0, bh1) MF(d10, ah1, bl0) MF(d11, ah1, bl1) MF(c10, ah1, bh0) MF(d10, al1, bh0) MF(c11, ah1, bh1)  \
  MF(d11, al1, bh1)
#define LT3_R1(A, o0, o1) LT_RD(ah0, A, o0) LT_RD(al0, A, o1)
#define LT3_R2(B, o0, o1, o2, o3) LT_RD(bl0, B, o1) LT_RD(bl1, B, o3)
#define LT3_R3(B, o0) LT_RD(bh0, B, o0)
#define LT3_R4(A, B, o2, o3) LT_RD(bh1, B, o2) LT_RD(ah1, A, o2) LT_RD(al1, A, o3)
// exact-fp16 activations (no lo plane): 8 MFMAs, 6 reads (after MFMA 4: ah0; after 6: bl0 bl1; after 7: bh0; after 8: bh1 ah1)
#define LT3_STEP2(MF, MID, R1, R2, R3, R4)                                                                     \
  LT_WAIT(4) MF(d00, ah0, bl0) LT_WAIT(2) MF(c00, ah0, bh0) MF(d01, ah0, bl1) LT_WAIT(1) MF(c01, ah0, bh1)       \
  MID R1 LT_WAIT(1) MF(d10, ah1, bl0) MF(d11, ah1, bl1) R2 MF(c10, ah1, bh0) R3 MF(c11, ah1, bh1) R4
#define LT3_STEP2_LAST(MF)                                                                                    \
  LT_WAIT(0) MF(d00, ah0, bl0) MF(c00, ah0, bh0) MF(d01, ah0, bl1) MF(c01, ah0, bh1) MF(d10, ah1, bl0)          \
  MF(d11, ah1, bl1) MF(c10, ah1, bh0) MF(c11, ah1, bh1)
#define LT3_BARRIER LT_WAIT(0) "s_barrier\n\t"

struct LtFrag3 { f32x4 ah0, ah1, al0, al1, bh0, bh1, bl0, bl1; };
#define LT3_OPERANDS                                                                                                 \
  : [c00] "+v"(acc[0][0]), [c01] "+v"(acc[0][1]), [c10] "+v"(acc[1][0]), [c11] "+v"(acc[1][1]),                       \
    [d00] "+v"(acc2[0][0]), [d01] "+v"(acc2[0][1]), [d10] "+v"(acc2[1][0]), [d11] "+v"(acc2[1][1]),                   \
    [ah0] "+v"(F.ah0), [ah1] "+v"(F.ah1), [al0] "+v"(F.al0), [al1] "+v"(F.al1), [bh0] "+v"(F.bh0), [bh1] "+v"(F.bh1), \
    [bl0] "+v"(F.bl0), [bl1] "+v"(F.bl1)                                                                             \
  : [a] "v"(a_cur), [b] "v"(b_cur), [an] "v"(a_nxt), [bn] "v"(b_nxt)                                                 \
  : "memory"

// fragments of a tile's first step (after the barrier that says its first stage has landed)
template <bool A16>
__device__ __forceinline__ void lt3_load_first(LtFrag3& F, unsigned a_cur, unsigned b_cur) {
  if constexpr (!A16)
    asm volatile(LT_RD(ah0, a, 0) LT_RD(bl0, b, 1024) LT_RD(bh0, b, 0) LT_RD(bl1, b, 5120) LT_RD(bh1, b, 4096)
                 LT_RD(al0, a, 1024) LT_RD(ah1, a, 4096) LT_RD(al1, a, 5120)
                 : [ah0] "=&v"(F.ah0), [ah1] "=&v"(F.ah1), [al0] "=&v"(F.al0), [al1] "=&v"(F.al1), [bh0] "=&v"(F.bh0),
                   [bh1] "=&v"(F.bh1), [bl0] "=&v"(F.bl0), [bl1] "=&v"(F.bl1)
                 : [a] "v"(a_cur), [b] "v"(b_cur)
                 : "memory");
  else
    asm volatile(LT_RD(ah0, a, 0) LT_RD(bl0, b, 1024) LT_RD(bh0, b, 0) LT_RD(bl1, b, 5120) LT_RD(bh1, b, 4096)
                 LT_RD(ah1, a, 2048)
                 : [ah0] "=&v"(F.ah0), [ah1] "=&v"(F.ah1), [bh0] "=&v"(F.bh0), [bh1] "=&v"(F.bh1), [bl0] "=&v"(F.bl0),
                   [bl1] "=&v"(F.bl1)
                 : [a] "v"(a_cur), [b] "v"(b_cur)
                 : "memory");
}
// One stage (two k16 steps).  On entry F holds the fragments of the stage's first step (reads possibly still in
// flight: the first waits are written for the steady-state order, which lt3_load_first also follows closely enough --
// its reads are all older than any wait here needs).  LAST: the tile's last stage -- no barrier, nothing reloaded in
// the second step; ends with the wait states an 8-pass MFMA result needs before other instructions touch it.
template <bool A16, bool TR, bool LAST>
__device__ __forceinline__ void lt3_stage(f32x16 (&acc)[2][2], f32x16 (&acc2)[2][2], LtFrag3& F, unsigned a_cur,
                                          unsigned b_cur, unsigned a_nxt, unsigned b_nxt) {
#define LT3_BODY3(MF)                                                                                                   \
  if constexpr (LAST)                                                                                                   \
    asm volatile(LT3_STEP3(MF, "", LT3_R1(a, 2048, 3072), LT3_R2(b, 2048, 3072, 6144, 7168), LT3_R3(b, 2048),          \
                           LT3_R4(a, b, 6144, 7168)) LT3_STEP3_LAST(MF) "s_nop 15\n\t" LT3_OPERANDS);                \
  else                                                                                                                  \
    asm volatile(LT3_STEP3(MF, "", LT3_R1(a, 2048, 3072), LT3_R2(b, 2048, 3072, 6144, 7168), LT3_R3(b, 2048),          \
                           LT3_R4(a, b, 6144, 7168))                                                                  \
                 LT3_STEP3(MF, LT3_BARRIER, LT3_R1(an, 0, 1024), LT3_R2(bn, 0, 1024, 4096, 5120), LT3_R3(bn, 0),       \
                           LT3_R4(an, bn, 4096, 5120)) LT3_OPERANDS)
#define LT3_BODY2(MF)                                                                                                   \
  if constexpr (LAST)                                                                                                   \
    asm volatile(LT3_STEP2(MF, "", LT_RD(ah0, a, 1024), LT_RD(bl0, b, 3072) LT_RD(bl1, b, 7168), LT_RD(bh0, b, 2048),  \
                           LT_RD(bh1, b, 6144) LT_RD(ah1, a, 3072)) LT3_STEP2_LAST(MF) "s_nop 15\n\t" LT3_OPERANDS);  \
  else                                                                                                                  \
    asm volatile(LT3_STEP2(MF, "", LT_RD(ah0, a, 1024), LT_RD(bl0, b, 3072) LT_RD(bl1, b, 7168), LT_RD(bh0, b, 2048),  \
                           LT_RD(bh1, b, 6144) LT_RD(ah1, a, 3072))                                                   \
                 LT3_STEP2(MF, LT3_BARRIER, LT_RD(ah0, an, 0), LT_RD(bl0, bn, 1024) LT_RD(bl1, bn, 5120),              \
                           LT_RD(bh0, bn, 0), LT_RD(bh1, bn, 4096) LT_RD(ah1, an, 2048)) LT3_OPERANDS)
  if constexpr (!A16) {
    if constexpr (TR) { LT3_BODY3(LT_MF_T); } else { LT3_BODY3(LT_MF_N); }
  } else {
    if constexpr (TR) { LT3_BODY2(LT_MF_T); } else { LT3_BODY2(LT_MF_N); }
  }
#undef LT3_BODY3
#undef LT3_BODY2
}

template <int EPI, int CLS, bool A16, int NST>
__global__ __launch_bounds__(768, 3) void gc_gemm_lt3_kernel(LtArgs g) {
  constexpr int PA = A16 ? 1 : 2, SPB = 2;
  constexpr int A_TILE = SPB * PA * 1024;       // bytes of one row tile per stage
  constexpr int B_TILE = SPB * 2 * 1024;        // bytes of one column tile per stage
  constexpr int A_STAGE = 8 * A_TILE, STAGE = A_STAGE + 4 * B_TILE;
  constexpr int GPW = (2 * A_TILE + B_TILE) / 1024;   // copy instructions per producer wave per stage
  constexpr int PF = NST - 1;
  constexpr bool TR = EPI != LT_EPI_F32;   // (AF16, H16, QKV epilogues)
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  // persistent tile walk, as generation 2, over 256-row tiles
  const int n_rt = (g.rows + 255) >> 8, n_ct = g.n >> 7;
  const int xcd = blockIdx.x & 7, wslot = blockIdx.x >> 3, wpx = gridDim.x >> 3;
  const int gxi = xcd % g.gx, gyi = xcd / g.gx, GY = 8 / g.gx;
  const int n_rz = n_rt * g.splits;
  const int rz_base = n_rz / GY, rz_extra = n_rz - rz_base * GY;
  const int rz_cnt = rz_base + (gyi < rz_extra ? 1 : 0);
  const int rz_first = gyi * rz_base + (gyi < rz_extra ? gyi : rz_extra);
  const int cpg = n_ct / g.gx;
  const int n_list = rz_cnt * cpg;
  const int n_mine = wslot < n_list ? (n_list - wslot + wpx - 1) / wpx : 0;
  if (n_mine == 0) return;
  const int nst = g.k_steps / SPB;
  auto tile_of = [&](int k, int& rt, int& ct, int& z) __attribute__((always_inline)) {
    const int j = wslot + k * wpx;
    const int jr = j / cpg;
    const int rz = rz_first + jr;
    ct = gxi * cpg + (j - jr * cpg);
    rt = rz / g.splits;
    z = rz - rt * g.splits;
  };

  if (wave >= 8) {
    // ---- producer p: row tiles 8 rt + 2 p, + 1 of A and column tile 4 ct + p of B, every stage of every tile
    const int p = wave - 8;
    const int a_dst = 2 * p * A_TILE, b_dst = A_STAGE + p * B_TILE;
    const int total = n_mine * nst;
    int ik = 0, ist = 0, ibuf = 0;
    const char* a_src = nullptr;
    const char* b_src = nullptr;
    size_t a_tile_stride = (size_t)g.a_steps * (PA * 1024);
    auto set_tile = [&](int k) __attribute__((always_inline)) {
      int rt, ct, z;
      tile_of(k, rt, ct, z);
      const int k0 = z * g.k_steps;
      a_src = reinterpret_cast<const char*>(g.a) + ((size_t)(rt * 8 + 2 * p) * g.a_steps + k0) * (PA * 1024) + lane * 16;
      b_src = reinterpret_cast<const char*>(g.wt) + ((size_t)(ct * 4 + p) * g.w_steps + k0) * 2048 + lane * 16;
    };
    auto issue_next = [&]() __attribute__((always_inline)) {
      char* base = lds + ibuf * STAGE;
      if (!(g.dbg & 2)) {                       // (diagnostic bit 1: no copies -- the consumers multiply what the LDS holds)
#pragma unroll
      for (int q = 0; q < A_TILE / 1024; ++q) glds16(a_src + q * 1024, base + a_dst + q * 1024);
#pragma unroll
      for (int q = 0; q < A_TILE / 1024; ++q) glds16(a_src + a_tile_stride + q * 1024, base + a_dst + A_TILE + q * 1024);
#pragma unroll
      for (int q = 0; q < B_TILE / 1024; ++q) glds16(b_src + q * 1024, base + b_dst + q * 1024);
      }
      a_src += A_TILE;
      b_src += B_TILE;
      ibuf = ibuf + 1 == NST ? 0 : ibuf + 1;
      if (++ist == nst) {
        ist = 0;
        if (++ik < n_mine) set_tile(ik);
      }
    };
    set_tile(0);
    int issued = 0;
    for (; issued < PF && issued < total; ++issued) issue_next();
    for (int s = 0; s < total; ++s) {
      const int later = issued - 1 - s;
      if (later >= PF - 1) wait_vm_c<(PF - 1) * GPW>();
      else wait_vm(later * GPW);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (issued < total) {
        issue_next();
        ++issued;
      }
    }
    return;
  }

  // ---- consumer (wm, wn): the 64 x 64 sub-tile of row tiles 2 wm + i (wm = 0..3), column tiles 2 wn + j
  const int wm = wave >> 1, wn = wave & 1;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const unsigned a_frag = lds0 + (wm * 2) * A_TILE + lane * 16;
  const unsigned b_frag = lds0 + A_STAGE + (wn * 2) * B_TILE + lane * 16;
  int buf = 0;
  for (int k = 0; k < n_mine; ++k) {
    int rt, ct, z;
    tile_of(k, rt, ct, z);
    f32x16 acc[2][2], acc2[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          acc[i][j][q] = 0.f;
          acc2[i][j][q] = 0.f;
        }
    LtFrag3 F;
    __builtin_amdgcn_s_barrier();               // the tile's first stage has landed
    asm volatile("" ::: "memory");
    lt3_load_first<A16>(F, a_frag + buf * STAGE, b_frag + buf * STAGE);
    for (int it = 0; it < nst - 1; ++it) {
      const int nbuf = buf + 1 == NST ? 0 : buf + 1;
      if (g.dbg & 4) {                          // (diagnostic bit 2: no fragment reads, no MFMAs -- the copy rate alone)
        __builtin_amdgcn_s_barrier();
      } else
      lt3_stage<A16, TR, false>(acc, acc2, F, a_frag + buf * STAGE, b_frag + buf * STAGE, a_frag + nbuf * STAGE,
                                b_frag + nbuf * STAGE);
      buf = nbuf;
    }
    lt3_stage<A16, TR, true>(acc, acc2, F, a_frag + buf * STAGE, b_frag + buf * STAGE, 0, 0);
    buf = buf + 1 == NST ? 0 : buf + 1;
    lt_epilogue<EPI, A16>(g, acc, acc2, rt * 256 + wm * 64, ct * 128 + wn * 64, z, lane);
  }
}

// tile / pipeline shapes (LtArgs.shape; 0 = the default)
template <int EPI, int CLS, bool A16, int WM, int SPB, int NST, int OCC>
static hipError_t launch_k(hipStream_t s, const LtArgs& g) {
  constexpr int lds = NST * SPB * (2 * WM * (A16 ? 1 : 2) + 8) * 1024;
  static_assert(lds * (OCC * 2 / WM) <= 160 * 1024, "LDS of the workgroups of one CU (OCC = waves per SIMD)");
  auto fn = gc_gemm_lt_kernel<EPI, CLS, A16, WM, SPB, NST, OCC>;
  // the dynamic-LDS limit is a per-DEVICE property of a kernel, raised once per (instantiation, device) whatever
  // thread gets here first (a thread that captures a HIP graph must find nothing lazy left to do)
  static std::atomic<unsigned long long> primed{0};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(primed.load(std::memory_order_acquire) & bit)) {
    e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    primed.fetch_or(bit, std::memory_order_release);
  }
  const int BM = 64 * WM;
  const int n_ct = g.n / 128, n_rt = (g.rows + BM - 1) / BM;
  const int GY = 8 / g.gx, n_rz = n_rt * g.splits;
  const int per_xcd = ((n_rz + GY - 1) / GY) * (n_ct / g.gx);
  hipLaunchKernelGGL(fn, dim3(8 * per_xcd), dim3(128 * WM), lds, s, g);
  return hipGetLastError();
}

template <int EPI, int CLS, bool A16, int NST, bool STAMP = false>
static hipError_t launch_k2(hipStream_t s, const LtArgs& g) {
  constexpr int lds = NST * 2 * (4 * (A16 ? 1 : 2) + 8) * 1024;
  static_assert(lds <= 160 * 1024, "one workgroup per CU");
  auto fn = gc_gemm_lt2_kernel<EPI, CLS, A16, NST, STAMP>;
  // the dynamic-LDS limit is a per-DEVICE property of a kernel, raised once per (instantiation, device) whatever
  // thread gets here first (a thread that captures a HIP graph must find nothing lazy left to do)
  static std::atomic<unsigned long long> primed{0};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(primed.load(std::memory_order_acquire) & bit)) {
    e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    primed.fetch_or(bit, std::memory_order_release);
  }
  const int n_ct = g.n / 128, n_rt = (g.rows + 127) / 128;
  const int GY = 8 / g.gx, n_rz = n_rt * g.splits;
  const int per_xcd = ((n_rz + GY - 1) / GY) * (n_ct / g.gx);
  static int cus = 0;                            // one workgroup per CU (each holds 96-160 KB of LDS)
  if (!cus) {
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
    cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const int wpx = per_xcd < cus / 8 ? per_xcd : cus / 8;
  hipLaunchKernelGGL(fn, dim3(8 * wpx), dim3(512), lds, s, g);
  return hipGetLastError();
}

template <int EPI, int CLS, bool A16, int NST>
static hipError_t launch_k3(hipStream_t s, const LtArgs& g) {
  constexpr int lds = NST * 2 * (8 * (A16 ? 1 : 2) + 8) * 1024;
  static_assert(lds <= 160 * 1024, "one workgroup per CU");
  auto fn = gc_gemm_lt3_kernel<EPI, CLS, A16, NST>;
  // the dynamic-LDS limit is a per-DEVICE property of a kernel, raised once per (instantiation, device) whatever
  // thread gets here first (a thread that captures a HIP graph must find nothing lazy left to do)
  static std::atomic<unsigned long long> primed{0};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(primed.load(std::memory_order_acquire) & bit)) {
    e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    primed.fetch_or(bit, std::memory_order_release);
  }
  const int n_ct = g.n / 128, n_rt = (g.rows + 255) / 256;
  const int GY = 8 / g.gx, n_rz = n_rt * g.splits;
  const int per_xcd = ((n_rz + GY - 1) / GY) * (n_ct / g.gx);
  static int cus = 0;
  if (!cus) {
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
    cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const int wpx = per_xcd < cus / 8 ? per_xcd : cus / 8;
  hipLaunchKernelGGL(fn, dim3(8 * wpx), dim3(768), lds, s, g);
  return hipGetLastError();
}

// LtArgs.shape: 0 / 1 = generation 1 (128 x 128, all waves copy and multiply, two workgroups per CU); 9 = generation 3
// (256 x 128, persistent, producer / consumer waves); 6 = generation 2 (128 x 128, persistent, producer / consumer) and
// 16 = its stamped instantiation, kept for tools/bench_gemm_lt.cpp
template <int EPI, int CLS, bool A16>
static hipError_t launch_t(hipStream_t s, const LtArgs& g) {
  switch (g.shape) {
    case 0:
    case 1: return launch_k<EPI, CLS, A16, 2, 2, 2, 2>(s, g);
    case 9: return launch_k3<EPI, CLS, A16, 3>(s, g);
    case 6: return launch_k2<EPI, CLS, A16, 4>(s, g);
    case 16:
      if constexpr (EPI == LT_EPI_AF16 && !A16) return launch_k2<EPI, CLS, A16, 4, true>(s, g);
      return hipErrorInvalidValue;
    default: return hipErrorInvalidValue;
  }
}

// the (class, epilogue) pairs the forward pass uses -- nothing else is instantiated
template <int CLS>
static hipError_t launch_c(hipStream_t s, const LtArgs& g, int epi, bool a16) {
#define GC_LT(EPI_) return a16 ? launch_t<EPI_, CLS, true>(s, g) : launch_t<EPI_, CLS, false>(s, g)
  if constexpr (CLS == gc::KC_GEMM_QKV) {
    if (epi == LT_EPI_QKV) { GC_LT(LT_EPI_QKV); }
  } else if constexpr (CLS == gc::KC_GEMM_FFW1) {
    if (epi == LT_EPI_AF16) { GC_LT(LT_EPI_AF16); }
    if (epi == LT_EPI_H16) return a16 ? launch_t<LT_EPI_H16, CLS, true>(s, g) : hipErrorInvalidValue;
  } else {
    if (epi == LT_EPI_F32) { GC_LT(LT_EPI_F32); }
  }
  return hipErrorInvalidValue;
#undef GC_LT
}

hipError_t launch_gemm_lt(hipStream_t s, int cls, const LtArgs& g_in, int epi, bool a16) {
  LtArgs g = g_in;
  if (g.rows <= 0) return hipSuccess;
  if (g.n <= 0 || g.n % 128 || g.k_steps < 2 || g.k_steps % 2 || g.splits < 1 || !g.a || !g.wt || !g.out)
    return hipErrorInvalidValue;
  if ((size_t)g.splits * g.k_steps > (size_t)g.a_steps || (size_t)g.splits * g.k_steps > (size_t)g.w_steps)
    return hipErrorInvalidValue;
  if (epi == LT_EPI_AF16 && (g.splits != 1 || g.out_steps * 16 < g.n)) return hipErrorInvalidValue;
  if (epi == LT_EPI_QKV && (g.splits != 1 || !g.kv16 || g.kv_d % 32 || g.n != 3 * g.kv_d || g.bias))
    return hipErrorInvalidValue;
  const int n_ct = g.n / 128;
  if (g.gx != 1 && g.gx != 2 && g.gx != 4 && g.gx != 8) {
    // default: as many column groups as divide the column tiles (a group's weight blocks then stay in one 4-MB L2)
    g.gx = 1;
    for (int c = 8; c >= 2; c >>= 1)
      if (n_ct % c == 0) { g.gx = c; break; }
  }
  if (n_ct % g.gx) return hipErrorInvalidValue;
  switch (cls) {
    case gc::KC_GEMM_QKV: return launch_c<gc::KC_GEMM_QKV>(s, g, epi, a16);
    case gc::KC_GEMM_FFW1: return launch_c<gc::KC_GEMM_FFW1>(s, g, epi, a16);
    case gc::KC_GEMM_FFW2: return launch_c<gc::KC_GEMM_FFW2>(s, g, epi, a16);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace gc_lt
