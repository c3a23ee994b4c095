// gfx950 (MI355X / CDNA4) kernels of the GenCast denoiser + DPM-Solver++2S path.
//
// All arithmetic is float32.  Dense projections run on the exact-f32 matrix
// cores (v_mfma_f32_32x32x2_f32: bit-for-bit an f32 fmaf chain, 64 FLOP/clk/SIMD,
// MI355X_MICROARCH.md "Matrix cores"), so parity with the f32/f64 oracle is
// limited only by summation order.
//
// MFMA operand convention used everywhere below (cdna_hip_programming.md §3):
//   D[32x32] += A[32x2] * B[2x32];  lane l supplies A[i = l&31][k = l>>5] and
//   B[k = l>>5][j = l&31];  D register g of lane l is D[row = (g&3)+8*(g>>2)+4*(l>>5)][col = l&31].
// A dot product does not care in which order k is visited, so within a 16-wide
// K chunk lane-half hh = l>>5 walks k = hh*8 + j (j = 0..7): every lane then
// reads 8 CONTIGUOUS floats of its row (two 16-byte loads) for both operands.
// Weights are stored transposed ([out][in]) so the B operand has the same
// row-contiguous shape as A.
#include "gc_kernels.h"

#include <algorithm>
#include <atomic>
#include <type_traits>

#include <math.h>
#include <stdlib.h>

// This file is compiled TWICE (csrc/build.sh): as it is -> namespace gc; with -DGC_TU_A16 -> namespace gc_a16,
// where the weight-streaming GEMM / MLP / FFW / out-projection launchers instantiate their kernels with
// A16 = true ("fp16 node features": the activation operand is exact fp16, 2 MFMAs per product).  Everything
// else is simply compiled again under the other namespace and never called.
#ifdef GC_TU_A16
#define GC_TU_NS gc_a16
#else
#define GC_TU_NS gc
#endif
namespace GC_TU_NS {
#ifdef GC_TU_A16
constexpr bool kTuA16 = true;
#else
constexpr bool kTuA16 = false;
#endif

#include "gc_dev_common.inc"


// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: one process may
// drive several GPUs from several threads (one handle each), so "already raised" is tracked per
// device id, lock-free (setting it twice is harmless).
struct DynLdsOnce {
  std::atomic<unsigned long long> done{0};
  hipError_t ensure(const void* fn, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
  }
};


// acc[nt] += A[32 x K] * W[K x 32] for NT column tiles.
//   a_row : this lane's A row, already offset by hh*8  (LDS or global)
//   w_row : this lane's W^T row of column tile 0, already offset by hh*8 (global)
//   w_tile_stride : floats between consecutive 32-column tiles of W^T
template <int NT>
__device__ __forceinline__ void wave_gemm(f32x16 (&acc)[NT], const float* __restrict__ a_row,
                                          const float* __restrict__ w_row, size_t w_tile_stride,
                                          int K) {
  f32x4 bc[NT][2];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    bc[nt][0] = ld4(w_row + nt * w_tile_stride);
    bc[nt][1] = ld4(w_row + nt * w_tile_stride + 4);
  }
  for (int k0 = 0; k0 < K; k0 += 16) {
    const f32x4 a0 = ld4(a_row + k0);
    const f32x4 a1 = ld4(a_row + k0 + 4);
    const int kn = (k0 + 16 < K) ? k0 + 16 : k0;  // prefetch next chunk (re-reads the last one)
    f32x4 bn[NT][2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      bn[nt][0] = ld4(w_row + nt * w_tile_stride + kn);
      bn[nt][1] = ld4(w_row + nt * w_tile_stride + kn + 4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma32(a0[j], bc[nt][0][j], acc[nt]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma32(a1[j], bc[nt][1][j], acc[nt]);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      bc[nt][0] = bn[nt][0];
      bc[nt][1] = bn[nt][1];
    }
  }
}

// ----------------------------------------------------------------------------
// gc_cond: noise-level encoder + every conditioning vector in one launch.
// Reference: FourierFeaturesMLP (common/mlp.py:255-265; model_utils.py:728-757)
// and LinearNormConditioning's linear layer for all 42 live sites (mlp.py:59-65).
// cond_out[b][j] = bc_all[j] + sum_i c[b][i] * wc_all[i][j]; the "+1" of the
// scale halves is folded into bc_all on the host.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gc_cond_kernel(
    const float* __restrict__ sigma_dev, float sigma_scalar, int B, const float* __restrict__ w0t,
    const float* __restrict__ b0, const float* __restrict__ w1t, const float* __restrict__ b1,
    int nfreq, int nhid, float base_period, const float* __restrict__ wc_all,
    const float* __restrict__ bc_all, int total, float* __restrict__ cond_vec,
    float* __restrict__ cond_out) {
  __shared__ float feats[256];
  __shared__ float hid[128];
  __shared__ float cvec[kCondDim];
  const int tid = threadIdx.x;
  const int j = blockIdx.x * 256 + tid;
  for (int b = 0; b < B; ++b) {
    const float sigma = sigma_dev ? sigma_dev[b] : sigma_scalar;
    const float x = logf(sigma);
    if (tid < 2 * nfreq) {
      const int k = (tid < nfreq) ? tid : tid - nfreq;
      const float w = (float)(2.0 * M_PI * (double)(k + 1) / (double)base_period);
      const float ang = x * w;
      feats[tid] = (tid < nfreq) ? cosf(ang) : sinf(ang);
    }
    __syncthreads();
    if (tid < nhid) {
      float s = b0[tid];
      for (int i = 0; i < 2 * nfreq; ++i) s += feats[i] * w0t[tid * 2 * nfreq + i];
      hid[tid] = gelu_tanh(s);
    }
    __syncthreads();
    if (tid < kCondDim) {
      float s = b1[tid];
      for (int i = 0; i < nhid; ++i) s += hid[i] * w1t[tid * nhid + i];
      cvec[tid] = s;
      if (blockIdx.x == 0) cond_vec[b * kCondDim + tid] = s;
    }
    __syncthreads();
    if (j < total) {
      float s = bc_all[j];
#pragma unroll
      for (int i = 0; i < kCondDim; ++i) s += cvec[i] * wc_all[(size_t)i * total + j];
      cond_out[(size_t)b * total + j] = s;
    }
    __syncthreads();
  }
}

// The sampler's form: the conditioning of EVERY denoiser call of a sample in one launch (the noise levels
// are known before the loop starts).  blockIdx.y = call; every batch element shares the call's sigma.
// cond_out[call][b][j].
__global__ __launch_bounds__(256) void gc_cond_multi_kernel(
    SigmaList sl, int B, const float* __restrict__ w0t, const float* __restrict__ b0,
    const float* __restrict__ w1t, const float* __restrict__ b1, int nfreq, int nhid, float base_period,
    const float* __restrict__ wc_all, const float* __restrict__ bc_all, int total, float* __restrict__ cond_out) {
  __shared__ float feats[256];
  __shared__ float hid[128];
  __shared__ float cvec[kCondDim];
  const int tid = threadIdx.x, call = blockIdx.y;
  const int j = blockIdx.x * 256 + tid;
  const float x = logf(sl.v[call]);
  if (tid < 2 * nfreq) {
    const int k = (tid < nfreq) ? tid : tid - nfreq;
    const float w = (float)(2.0 * M_PI * (double)(k + 1) / (double)base_period);
    const float ang = x * w;
    feats[tid] = (tid < nfreq) ? cosf(ang) : sinf(ang);
  }
  __syncthreads();
  if (tid < nhid) {
    float a = b0[tid];
    for (int i = 0; i < 2 * nfreq; ++i) a += feats[i] * w0t[tid * 2 * nfreq + i];
    hid[tid] = gelu_tanh(a);
  }
  __syncthreads();
  if (tid < kCondDim) {
    float a = b1[tid];
    for (int i = 0; i < nhid; ++i) a += hid[i] * w1t[tid * nhid + i];
    cvec[tid] = a;
  }
  __syncthreads();
  if (j < total) {
    float a = bc_all[j];
#pragma unroll
    for (int i = 0; i < kCondDim; ++i) a += cvec[i] * wc_all[(size_t)i * total + j];
    for (int b = 0; b < B; ++b) cond_out[((size_t)call * B + b) * total + j] = a;
  }
}

hipError_t launch_cond_multi(hipStream_t s, const SigmaList& sl, int ncalls, int B, const float* w0t,
                             const float* b0, const float* w1t, const float* b1, int nfreq, int nhid,
                             float base_period, const float* wc_all, const float* bc_all, int total,
                             float* cond_out) {
  if (ncalls < 1 || ncalls > kMaxSigmaList) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gc_cond_multi_kernel, dim3((total + 255) / 256, ncalls), dim3(256), 0, s, sl, B, w0t, b0, w1t,
                     b1, nfreq, nhid, base_period, wc_all, bc_all, total, cond_out);
  return hipGetLastError();
}

hipError_t launch_cond(hipStream_t s, const float* sigma_dev, float sigma_scalar, int B,
                       const float* w0t, const float* b0, const float* w1t, const float* b1,
                       int nfreq, int nhid, float base_period, const float* wc_all,
                       const float* bc_all, int total, float* cond_vec, float* cond_out) {
  const int grid = (total + 255) / 256;
  hipLaunchKernelGGL(gc_cond_kernel, dim3(grid), dim3(256), 0, s, sigma_dev, sigma_scalar, B, w0t, b0,
                     w1t, b1, nfreq, nhid, base_period, wc_all, bc_all, total, cond_vec, cond_out);
  return hipGetLastError();
}

// ----------------------------------------------------------------------------
// gc_mlp: fused MLPWithNormConditioning on a 32-row tile (common/mlp.py:115-147):
//   concat(segments) -> Linear -> swish -> Linear -> [LayerNorm] -> [cond] -> [+res]
// Segments implement the InteractionNetwork gathers (typed_graph_net.py:134-159,
// 295-326): edge MLP input [e | n_s[senders] | n_r[receivers]], node MLP input
// [n | sum of received edges].  The hidden activation never leaves LDS.
// 4 waves split the output columns; each keeps NT 32x32 accumulators.
// ----------------------------------------------------------------------------
constexpr int kMlpBK = 32;             // K tile staged per step
constexpr int kMlpLd = kMlpBK + 4;     // 36 floats = 9 x 16 B: conflict-free ds_read_b128 / ds_write_b128

// One 32-wide K step of a [32 x K] x [K x (4*NT*32)] product: the A fragment of this lane's row is
// read from `a_lds` (row-major, leading dimension lda), the W^T tile from `w_lds` ([cols][36]).
template <int NT, bool F16>
__device__ __forceinline__ void mlp_tile_mfma(f32x16 (&acc)[NT], f32x16 (&acc2)[NT], const float* a_lds_row,
                                              const float* w_lds_row) {
  // a_lds_row / w_lds_row point at this lane's row of a 32-wide K tile (no lane-half offset applied)
  const int hh = (threadIdx.x & 63) >> 5;
  if constexpr (!F16) {
    f32x4 a4[4], b4[NT][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) a4[q] = ld4(a_lds_row + hh * 16 + 4 * q);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) b4[nt][q] = ld4(w_lds_row + nt * 32 * kMlpLd + hh * 16 + 4 * q);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma32(a4[q][e], b4[nt][q][e], acc[nt]);
  } else {
    // S16 tiles: [hi: 32 halfs | lo: 32 halfs] per row; k-step ks covers halfs 16ks + 8hh ..
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const f32x4 ah = ld4(a_lds_row + ks * 8 + hh * 4);
      const f32x4 al = ld4(a_lds_row + 16 + ks * 8 + hh * 4);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const f32x4 bh = ld4(w_lds_row + nt * 32 * kMlpLd + ks * 8 + hh * 4);
        const f32x4 bl = ld4(w_lds_row + nt * 32 * kMlpLd + 16 + ks * 8 + hh * 4);
        acc[nt] = mfma16(ah, bh, acc[nt]);
        acc2[nt] = mfma16(ah, bl, acc2[nt]);
        acc2[nt] = mfma16(al, bh, acc2[nt]);
      }
    }
  }
}

// WM = 1: 32-row tile, 4 waves; WM = 2: 64-row tile, 8 waves (halves the weight-tile traffic per
// row, which is what bounds the big edge MLPs once the products run at fp16 MFMA speed).
template <int NT1, int NT2, bool F16, int WM>
__global__ __launch_bounds__(256 * WM) void gc_mlp_kernel(MlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int HID = NT1 * 128, NPAD = NT2 * 128;
  constexpr int WROWS = (HID > NPAD) ? HID : NPAD;
  constexpr int LDH = HID + 4, LDY = NPAD + 4;
  constexpr int BM = 32 * WM, NTHR = 256 * WM, RPP = NTHR / 8;
  static_assert(HID % RPP == 0 && NPAD % RPP == 0 && NPAD <= HID, "unsupported MLP shape");
  float* Wbuf = smem;                          // [WROWS][36] staged W^T tile
  float* Abuf = smem + WROWS * kMlpLd;         // [BM][36] staged (gathered) input tile
  float* Hbuf = Abuf + BM * kMlpLd;            // [BM][HID+4] hidden activations; later the output tile

  const int tid = threadIdx.x;
  const int wave_all = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int wm = wave_all >> 2, wave = wave_all & 3;   // row half, column quarter
  const int row0 = blockIdx.x * BM;
  const int lrow = tid >> 3, lc4 = tid & 7;    // staging role: row lrow (+RPP i), 16-byte piece lc4

  // this thread's source row for every segment (fixed for the whole kernel)
  int grow = row0 + lrow;
  if (grow >= a.rows) grow = a.rows - 1;
  const int item = grow / a.B, bidx = grow - item * a.B;
  // Per-segment source row of this thread.  Kept in NAMED scalars (never arrays indexed at run
  // time: hipcc would demote those to scratch memory and every K tile would pay a scratch load).
  const float *srow0 = nullptr, *srow1 = nullptr, *srow2 = nullptr;
  const float *saff0 = nullptr, *saff1 = nullptr, *saff2 = nullptr;
  int swid0 = 0, swid1 = 0, swid2 = 0;
  {
    auto src_of = [&](const Segment& sg, const float*& rowp, const float*& affp, int& wid) {
      size_t srow = sg.index ? (size_t)sg.index[item] : (size_t)item;
      if (!sg.bcast) srow = srow * a.B + bidx;
      rowp = sg.ptr + srow * sg.ld + lc4 * 4;
      affp = sg.affine ? sg.affine + (size_t)bidx * a.cond_stride + lc4 * 4 : nullptr;
      wid = sg.width;
    };
    src_of(a.seg[0], srow0, saff0, swid0);
    if (a.nseg > 1) src_of(a.seg[1], srow1, saff1, swid1);
    if (a.nseg > 2) src_of(a.seg[2], srow2, saff2, swid2);
  }

  // ---------------- phase 1: hidden = swish(concat(segments) @ W1 + b1) ----------------
  f32x16 acc[NT1], accx[NT1];
#pragma unroll
  for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      acc[nt][q] = 0.f;
      accx[nt][q] = 0.f;
    }
  {
    const int nk_total = (swid0 + swid1 + swid2) / kMlpBK;
    constexpr int WL = HID / RPP;              // 16-byte W pieces per thread per tile
    f32x4 ra, rw[WL];
    int sidx = 0, kin = 0;                     // loader position: segment, k offset inside it
    const float* cur_row = srow0;
    const float* cur_aff = saff0;
    int cur_wid = swid0;
    auto load_tile = [&](int ktile) {
      f32x4 v = ld4(cur_row + kin);
      if (cur_aff) {
        const f32x4 sc = ld4(cur_aff + kin), of = ld4(cur_aff + cur_wid + kin);
        v = v * sc + of;
      }
      ra = r16_if(v, a.round16);
      const float* wp = a.w1t + (size_t)lrow * a.ldw1 + ktile * kMlpBK + lc4 * 4;
#pragma unroll
      for (int i = 0; i < WL; ++i) rw[i] = ld4(wp + (size_t)(RPP * i) * a.ldw1);
      kin += kMlpBK;
      if (kin >= cur_wid) {                    // next segment
        kin = 0;
        ++sidx;
        cur_row = (sidx == 1) ? srow1 : srow2;
        cur_aff = (sidx == 1) ? saff1 : saff2;
        cur_wid = (sidx == 1) ? swid1 : swid2;
      }
    };
    load_tile(0);
    for (int kt = 0; kt < nk_total; ++kt) {
      if (kt) __syncthreads();                 // everyone finished reading the previous tile
      if constexpr (F16) {                     // split the f32 input on the fly into the S16 tile
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        _Float16 hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split16(ra[e], hi[e], lo[e]);
        _Float16* ap = reinterpret_cast<_Float16*>(Abuf + lrow * kMlpLd) + lc4 * 4;
        *reinterpret_cast<f16x4*>(ap) = f16x4{hi[0], hi[1], hi[2], hi[3]};
        *reinterpret_cast<f16x4*>(ap + 32) = f16x4{lo[0], lo[1], lo[2], lo[3]};
      } else
      st4(Abuf + lrow * kMlpLd + lc4 * 4, ra);
#pragma unroll
      for (int i = 0; i < WL; ++i) st4(Wbuf + (lrow + RPP * i) * kMlpLd + lc4 * 4, rw[i]);
      __syncthreads();
      if (kt + 1 < nk_total) load_tile(kt + 1);   // in flight during the MFMAs below
      mlp_tile_mfma<NT1, F16>(acc, accx, Abuf + (wm * 32 + r) * kMlpLd, Wbuf + (wave * NT1 * 32 + r) * kMlpLd);
    }
  }
  // hidden = swish(acc + b1 [+ pre-projected node terms]) -> LDS.  The node terms belong to an edge
  // MLP whose first layer is split by input block:
  //   concat([e, n_s, n_r]) @ W1 = e @ Wa + (n_s @ Wb)[senders] + (n_r @ Wc)[receivers];
  // the two per-node products are computed once per node by a plain GEMM and gathered here.
  float bias1[NT1];
#pragma unroll
  for (int nt = 0; nt < NT1; ++nt) bias1[nt] = a.b1[wave * NT1 * 32 + nt * 32 + r];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const float* add0 = nullptr;
    const float* add1 = nullptr;
    if (a.nadd > 0) {
      int arow = row0 + wm * 32 + acc_row(q, hh);
      if (arow >= a.rows) arow = a.rows - 1;
      const int it = arow / a.B, bb = arow - it * a.B;
      add0 = a.add[0].ptr + ((size_t)a.add[0].index[it] * a.B + bb) * HID;
      if (a.nadd > 1) add1 = a.add[1].ptr + ((size_t)a.add[1].index[it] * a.B + bb) * HID;
    }
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt) {
      const int col = wave * NT1 * 32 + nt * 32 + r;
      float v = acc[nt][q];
      if constexpr (F16) v = hilo(v, accx[nt][q]);
      if (add0) v += add0[col];
      if (add1) v += add1[col];
      v = r16_if(swish(v + bias1[nt]), a.round16);
      if constexpr (F16) store_s16(Hbuf, (size_t)(wm * 32 + acc_row(q, hh)), LDH, col, v);
      else Hbuf[(wm * 32 + acc_row(q, hh)) * LDH + col] = v;
    }
  }

  // ---------------- phase 2: y = hidden @ W2 + b2 ---------------------------------------------
  f32x16 acc2[NT2], acc2x[NT2];
#pragma unroll
  for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      acc2[nt][q] = 0.f;
      acc2x[nt][q] = 0.f;
    }
  {
    constexpr int WL2 = NPAD / RPP;
    constexpr int nk2 = HID / kMlpBK;
    f32x4 rw[WL2];
    auto load_tile2 = [&](int ktile) {
      const float* wp = a.w2t + (size_t)lrow * HID + ktile * kMlpBK + lc4 * 4;
#pragma unroll
      for (int i = 0; i < WL2; ++i) rw[i] = ld4(wp + (size_t)(RPP * i) * HID);
    };
    load_tile2(0);
    for (int kt = 0; kt < nk2; ++kt) {
      __syncthreads();                         // also orders the Hbuf writes before the first read
#pragma unroll
      for (int i = 0; i < WL2; ++i) st4(Wbuf + (lrow + RPP * i) * kMlpLd + lc4 * 4, rw[i]);
      __syncthreads();
      if (kt + 1 < nk2) load_tile2(kt + 1);
      mlp_tile_mfma<NT2, F16>(acc2, acc2x, Hbuf + (wm * 32 + r) * LDH + kt * kMlpBK,
                              Wbuf + (wave * NT2 * 32 + r) * kMlpLd);
    }
  }
  __syncthreads();                             // hidden tile no longer needed: reuse it for the output tile
  float* Ybuf = Hbuf;
#pragma unroll
  for (int nt = 0; nt < NT2; ++nt) {
    const int col = wave * NT2 * 32 + nt * 32 + r;
    const float bias = a.b2[col];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      float v = acc2[nt][q];
      if constexpr (F16) v = hilo(v, acc2x[nt][q]);
      Ybuf[(wm * 32 + acc_row(q, hh)) * LDY + col] = r16_if(v + bias, a.round16);
    }
  }
  __syncthreads();

  // epilogue: each wave finishes 8 rows; a lane owns columns lane + 64 j (j < NPAD/64).  All 8
  // rows are processed together -- residual / conditioning loads issued up front, the 16 row
  // statistics reduced side by side -- so no row waits on another row's memory round trip.
  const int n = a.n_out;
  const float inv_n = 1.0f / (float)n;
  constexpr int CPL = NPAD / 64;               // columns per lane
  float yv[8][CPL], rv[8][CPL];
  const int rbase = wave_all * 8;
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) {
    const int orow = row0 + rbase + rr;
    const bool live = orow < a.rows;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int c = lane + 64 * j;
      yv[rr][j] = (c < n) ? Ybuf[(rbase + rr) * LDY + c] : 0.f;
      rv[rr][j] = (a.residual && live && c < n) ? a.residual[(size_t)orow * n + c] : 0.f;
    }
  }
  float mean[8], rstd[8];
  if (a.do_ln) {
    float s1[8], s2[8];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      s1[rr] = 0.f;
      s2[rr] = 0.f;
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        s1[rr] += yv[rr][j];
        s2[rr] += yv[rr][j] * yv[rr][j];
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        s1[rr] += __shfl_xor(s1[rr], o);
        s2[rr] += __shfl_xor(s2[rr], o);
      }
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      mean[rr] = s1[rr] * inv_n;
      const float var = fmaxf(s2[rr] * inv_n - mean[rr] * mean[rr], 0.f);
      rstd[rr] = 1.0f / sqrtf(var + 1e-6f);
    }
  } else {
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      mean[rr] = 0.f;
      rstd[rr] = 1.f;
    }
  }
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) {
    const int orow = row0 + rbase + rr;
    if (orow >= a.rows) break;
    const float* cs = a.cond ? a.cond + (size_t)(orow % a.B) * a.cond_stride : nullptr;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int c = lane + 64 * j;
      if (c < n) {
        float v = (yv[rr][j] - mean[rr]) * rstd[rr];
        if (cs) v = v * cs[c] + cs[n + c];
        a.out[(size_t)orow * a.ldo + c] = r16_if(r16_if(v, a.round_out) + rv[rr][j], a.round_out);
      }
    }
  }
}

// ----------------------------------------------------------------------------
// gc_mlp_ws: the fused MLP in weight-streaming form (f16x3).  Same contract as gc_mlp_kernel, but
//   * W1 / W2 fragments are loaded straight from their WF16 images into MFMA registers through a
//     two-group register ring (no LDS staging, no barrier per K tile);
//   * the gathered input is staged in 128-wide K chunks, double-buffered in LDS: one barrier per
//     chunk, the next chunk's gather in flight during the MFMAs;
//   * a workgroup covers 32*MT rows: at MT = 2 every weight fragment feeds two row tiles, which
//     halves the L2 -> CU weight traffic that bounds the big edge MLPs (1 MB of weights per tile);
//   * both products are computed transposed (the weight fragment is the MFMA's A operand), so a
//     lane ends up with 4 consecutive columns of one row: the hidden tile goes to LDS with 8-byte
//     writes and the output tile with 16-byte writes.
// The A chunks, the hidden tile and the output tile share one LDS region (~67 KB: 2 per CU).
// ----------------------------------------------------------------------------
// i-th of three values.  Written with assignments, not `c ? x : y`: with lvalue operands the
// conditional operator is itself an lvalue, clang then selects between the operands' ADDRESSES and
// everything they live in (kernel-argument struct, lambda captures) is forced into scratch memory.
template <class T>
__device__ __forceinline__ T pick3(int i, T x0, T x1, T x2) {
  T v = x2;
  if (i == 1) v = x1;
  if (i == 0) v = x0;
  return v;
}

// Register ring of R k16-steps of weight fragments (NT column tiles each, hi and lo).
template <int NT, int R>
__device__ __forceinline__ void ws_ring_fill(f32x4 (&wh)[R][NT], f32x4 (&wl)[R][NT], const float* wf,
                                             size_t ct_stride, int steps_total) {
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int sn = j < steps_total ? j : steps_total - 1;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      wh[j][nt] = ld4(wf + nt * ct_stride + (size_t)sn * 512);
      wl[j][nt] = ld4(wf + nt * ct_stride + (size_t)sn * 512 + 256);
    }
  }
}

// Four k16-steps of  acc[mt][nt] += W_frag(nt) x A_rows(mt)^T  (transposed product: register q of
// lane (r, hh) = column acc_row(q, hh) of the nt tile, row r of the mt tile), consuming ring slots
// J0 .. J0+3 and refilling each, right after its MFMAs, with the step R ahead -- so R-1 steps of
// loads stay in flight behind the MFMAs (the scheduling barriers stop hipcc from sinking the
// refills to the end of the block, which exposes a full L2 round trip per block).
// `a_row`: this lane's LDS row (S16) of row tile 0, already offset by hh*4; kstep0 = k16 index of
// the first step inside that row; `s` = position in the weight stream (advanced by 4).
// A16: the A operand holds exact fp16 values (its lo plane is all zeros): the wh x al product is skipped.
// F32 (exact-f32 family, gc_set_option precision = f32): the LDS rows hold plain float32 (`a_row` offset by hh*8), the
// ring holds the WF32 image (the lane's 8 consecutive k of a k16 step as two 16-byte halves in wh / wl), and a k16
// step is 8 v_mfma_f32_32x32x2_f32 into ONE accumulator (acc2 is not touched) -- bit for bit an f32 fmaf chain.
template <int MT, int NT, int R, int J0, bool A16 = false, bool F32 = false>
__device__ __forceinline__ void ws_quad(f32x16 (&acc)[MT][NT], f32x16 (&acc2)[MT][NT], f32x4 (&wh)[R][NT],
                                        f32x4 (&wl)[R][NT], const float* a_row, int mt_stride, int kstep0,
                                        const float* wf, size_t ct_stride, int& s, int steps_total) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int kk = kstep0 + j;
    const int off = F32 ? kk * 16 : (kk >> 1) * 32 + (kk & 1) * 8;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if constexpr (F32) {
        const f32x4 a0 = ld4(a_row + mt * mt_stride + off), a1 = ld4(a_row + mt * mt_stride + off + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma32(wh[J0 + j][nt][e], a0[e], acc[mt][nt]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma32(wl[J0 + j][nt][e], a1[e], acc[mt][nt]);
        continue;
      }
      const f32x4 ah = ld4(a_row + mt * mt_stride + off);
      f32x4 al;
      if constexpr (!A16) al = ld4(a_row + mt * mt_stride + off + 16);
      // the two MFMAs into acc2 are kept apart (a dependent MFMA cannot issue back to back)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc2[mt][nt] = mfma16(wl[J0 + j][nt], ah, acc2[mt][nt]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma16(wh[J0 + j][nt], ah, acc[mt][nt]);
      if constexpr (!A16) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc2[mt][nt] = mfma16(wh[J0 + j][nt], al, acc2[mt][nt]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    int sn = s + R;
    if (sn >= steps_total) sn = steps_total - 1;   // clamped: a harmless re-read at the tail
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      wh[J0 + j][nt] = ld4(wf + nt * ct_stride + (size_t)sn * 512);
      wl[J0 + j][nt] = ld4(wf + nt * ct_stride + (size_t)sn * 512 + 256);
    }
    ++s;
    __builtin_amdgcn_sched_barrier(0);
  }
}

// LayerNorm arithmetic of the fused MLP's two row epilogues (one row per output row / one row per summed triple), written
// with EXPLICIT fused multiply-adds: with -ffp-contract=fast hipcc chose the contraction of `s2 * inv_n - mean * mean` per
// call site, and the two epilogues then disagreed in the last bit of rstd for some rows -- which broke the bit-identity of
// the fused mesh2grid sum with the two-launch form it replaces.
__device__ __forceinline__ void ln_accumulate(const f32x4& y, float& s1, float& s2) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    s1 += y[e];
    s2 = __builtin_fmaf(y[e], y[e], s2);
  }
}
__device__ __forceinline__ void ln_finish(float s1, float s2, float inv_n, float& mean, float& rstd) {
  mean = s1 * inv_n;
  const float var = fmaxf(__builtin_fmaf(-mean, mean, s2 * inv_n), 0.f);   // one-pass variance (flax LayerNorm), clipped at 0
  rstd = 1.0f / sqrtf(var + 1e-6f);
}
__device__ __forceinline__ f32x4 ln_affine(const f32x4& y, float mean, float rstd, const f32x4& sc, const f32x4& of) {
  f32x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = __builtin_fmaf((y[e] - mean) * rstd, sc[e], of[e]);
  return r;
}

// NWC waves split the hidden columns (32*NT1 each), NW2 <= NWC of them the output columns (32*NT2
// each): 4 / 4 up to hidden = 256; hidden = 512 runs 8 column waves with NT1 = 2, which keeps the
// accumulators at 64*MT registers and two waves per SIMD where NT1 = 4 allowed one.
template <int NT1, int WM, int NWC>
constexpr int mlp_ws_occ() { return NWC >= 8 ? (NT1 == 1 ? 4 : 2) : ((WM >= 2 || NT1 >= 4) ? 1 : 2); }
// A16: exact-fp16 staged inputs and hidden tile (fp16 node features): 2 MFMAs per product (see gc_gemm_ws_kernel)
// F32 (exact-f32 family): w1f / w2f are WF32 images, the staged input and the hidden tile are plain float32 in LDS,
// both products run on v_mfma_f32_32x32x2_f32 (ws_quad's F32 form).
// The kernel body, with the workgroup's tile index as a parameter: gc_mlp_ws_kernel runs it on blockIdx.x,
// gc_mlp_ws_pair_kernel runs TWO independent MLPs of the same shape in one launch (blocks [0, na) the first, the rest the
// second), so that their partly filled last rounds of workgroups share the chip.
template <int NT1, int NT2, int MT, int WM, int NWC, int NW2, int OCC, bool A16, bool F32>
__device__ __forceinline__ void gc_mlp_ws_body(const MlpArgs& a, const int bid) {
  static_assert(!(A16 && F32), "one or the other");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int HID = NT1 * 32 * NWC, NPAD = NT2 * 32 * NW2, BM = 32 * MT * WM, NTHR = 64 * NWC * WM;
  // K chunk of the gathered input resident in LDS (double-buffered), and weight ring depth.
  // WM = 2: eight waves as 2 row halves x 4 column quarters on a 128-row tile, one workgroup per
  // CU; the two row halves read the same weight fragments (the second read hits the CU's L1), so
  // the L2 -> CU weight traffic per row is a quarter of the 32-row tile's.
  constexpr int KC = BM >= 128 ? 64 : 128, LDA = KC + 4, LDH = HID + 4, LDY = NPAD + 4;
  constexpr int PPR = KC / 4;                  // 16-byte pieces per row per chunk
  constexpr int RSTEP = NTHR / PPR;            // rows covered by one pass of the workgroup's threads
  constexpr int AP = BM / RSTEP;               // pieces per thread per chunk
  // ring depth in k16 steps (8 x NT registers each): 8 where the register budget allows
  constexpr int R1 = (OCC < 4 && (NT1 * MT <= 2 || (NT1 >= 4 && MT == 1))) ? 8 : 4;
  constexpr int R2 = (OCC < 4 && (NT2 * MT <= 2 || (NT1 >= 4 && NT2 * MT <= 4))) ? 8 : 4;
  int* srcoff = reinterpret_cast<int*>(smem);  // [3][BM] element offset of each row's source row
  int* srcb = srcoff + 3 * BM;                 // [BM]    batch index of each row
  int* addoff = srcb + BM;                     // [2][BM] element offset of each row's add-term rows (a.nadd > 0)
  float* region = smem + 6 * BM;               // A chunks [2][BM][LDA] | hidden [BM][LDH] | output [BM][LDY]

  const int tid = threadIdx.x;
  const int wave_all = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int wave = wave_all % NWC, wrow = (wave_all / NWC) * (MT * 32);   // column group, first row of the row half
  const int row0 = bid * BM;
  // a.tri (mesh2grid edge update with the segment sum in its epilogue): rows come as TRIPLES -- local row 3 j + s is edge
  // 3 g + s of unit u = bid * TPB + j = (grid node g, batch element b); a tile holds TPB whole triples (its last
  // BM - 3 TPB rows are padding), and the epilogue writes ONE row per unit: the sum of the triple's three results.
  constexpr int TPB = BM / 3;
  const int tri = a.tri;
  const int unit0 = bid * TPB, units = a.rows / 3;
  const int w0 = a.seg[0].width, w1 = a.nseg > 1 ? a.seg[1].width : 0, w2 = a.nseg > 2 ? a.seg[2].width : 0;
  const int ktot = w0 + w1 + w2, kpad = a.k1f;

  // Segment fields are copied into named scalars first: a select between loads of a.seg[i].x gets
  // turned into a run-time index into the argument struct, which then lives in scratch memory.
  const float *p0 = a.seg[0].ptr, *p1 = a.seg[1].ptr, *p2 = a.seg[2].ptr;
  const float *f0 = a.seg[0].affine, *f1 = a.nseg > 1 ? a.seg[1].affine : nullptr,
              *f2 = a.nseg > 2 ? a.seg[2].affine : nullptr;
  {
    const int *ix0 = a.seg[0].index, *ix1 = a.seg[1].index, *ix2 = a.seg[2].index;
    const int bc0 = a.seg[0].bcast, bc1 = a.seg[1].bcast, bc2 = a.seg[2].bcast;
    const int ld0 = a.seg[0].ld, ld1 = a.seg[1].ld, ld2 = a.seg[2].ld;
    for (int idx = tid; idx < a.nseg * BM; idx += NTHR) {
      const int sgi = idx / BM, i = idx - sgi * BM;
      int grow = row0 + i;
      if (grow >= a.rows) grow = a.rows - 1;
      int item = grow / a.B, b = grow - item * a.B;
      if (tri) {                                 // (padding rows and units beyond the end repeat a valid row; never stored)
        const int j = i / 3;
        int u = unit0 + j;
        if (u >= units) u = units - 1;
        const int gi = u / a.B;
        b = u - gi * a.B;
        item = 3 * gi + (i - 3 * j);
      }
      const int* ix = pick3(sgi, ix0, ix1, ix2);
      const int bc = pick3(sgi, bc0, bc1, bc2);
      const int ld = pick3(sgi, ld0, ld1, ld2);
      int srow = ix ? ix[item] : item;
      if (!bc) srow = srow * a.B + b;
      srcoff[idx] = srow * ld;
      if (sgi == 0) srcb[i] = b;
    }
    // Edge MLP with its first layer split by input block (concat([e, n_s, n_r]) @ W1 = e @ Wa + (n_s @ Wb)[senders]
    // + (n_r @ Wc)[receivers]): the per-node products were computed once per NODE by a plain GEMM (float32
    // pre-activation terms, [nodes * B][HID]) and are gathered per edge row in the phase-1 epilogue below.
    const int* ax0 = a.add[0].index;
    const int* ax1 = a.add[1].index;
    for (int idx = tid; idx < a.nadd * BM; idx += NTHR) {
      const int t = idx / BM, i = idx - t * BM;
      int grow = row0 + i;
      if (grow >= a.rows) grow = a.rows - 1;
      int item = grow / a.B, b = grow - item * a.B;
      if (tri) {
        const int j = i / 3;
        int u = unit0 + j;
        if (u >= units) u = units - 1;
        const int gi = u / a.B;
        b = u - gi * a.B;
        item = 3 * gi + (i - 3 * j);
      }
      const int* ax = t ? ax1 : ax0;
      addoff[idx] = ((ax ? ax[item] : item) * a.B + b) * HID;   // no index: a per-row term
    }
  }
  // per-segment affine (scale, offset) sources; segments without one read the identity
  const float* sc0 = f0 ? f0 : a.ones;
  const float* of0 = f0 ? f0 + w0 : a.zeros;
  const int as0 = f0 ? a.cond_stride : 0;
  const float* sc1 = f1 ? f1 : a.ones;
  const float* of1 = f1 ? f1 + w1 : a.zeros;
  const int as1 = f1 ? a.cond_stride : 0;
  const float* sc2 = f2 ? f2 : a.ones;
  const float* of2 = f2 ? f2 + w2 : a.zeros;
  const int as2 = f2 ? a.cond_stride : 0;
  __syncthreads();

  // ---- gather loader: piece i of this thread = row tid / PPR + RSTEP i, 16-byte column tid % PPR ----
  const int c4 = tid % PPR, prow = tid / PPR;
  // A16 (physical fp16 storage): a segment stored as halfs comes in 16-byte pieces of EIGHT k values and goes to
  // LDS as it is (no affine, no conversion, no rounding: the producer rounded).  Only segment 0 can still be a
  // float32 source (a.seg0_f32: the packed grid input, the statically embedded edge latents with their
  // conditioning affine, the structural features).  A chunk lies inside ONE segment (host: with more than one
  // segment every width is a multiple of KC), so "halfs or floats" is decided per chunk, uniformly.
  constexpr int PPR8 = KC / 8, RSTEP8 = NTHR / PPR8, AP8 = BM / RSTEP8;
  static_assert(!A16 || (AP8 >= 1 && AP8 * RSTEP8 == BM && AP8 <= AP), "fp16 piece mapping");
  const int c8 = tid % PPR8, prow8 = tid / PPR8;
  auto chunk_is_h16 = [&](int c) __attribute__((always_inline)) {
    if constexpr (!A16) return false;
    else return !(a.seg0_f32 && c * KC < w0);
  };
  // load_chunk only ISSUES the loads (raw rows + this chunk's scale/offset); every use of the
  // values waits until stage_chunk one chunk later, so the gather stays in flight behind the MFMAs.
  f32x4 ra[AP], rsc, rof;
  const float *rscp = nullptr, *rofp = nullptr;
  int rastr = 0;
  bool rlive = false;
  auto load_chunk = [&](int c) __attribute__((always_inline)) {
    if constexpr (A16) {
      if (chunk_is_h16(c)) {
        const int kglob = c * KC + c8 * 8;
        rlive = kglob < ktot;
        const int kq = rlive ? kglob : 0;
        const int sgi = (kq >= w0) + (kq >= w0 + w1);
        const int kin = kq - pick3(sgi, 0, w0, w0 + w1);
        const _Float16* base = as_h16(pick3(sgi, p0, p1, p2)) + kin;
#pragma unroll
        for (int i = 0; i < AP8; ++i)
          ra[i] = ld4(reinterpret_cast<const float*>(base + srcoff[sgi * BM + prow8 + RSTEP8 * i]));
        return;
      }
    }
    const int kglob = c * KC + c4 * 4;
    rlive = kglob < ktot;                      // beyond the real K: staged as zeros (W1 is zero-padded too)
    const int kq = rlive ? kglob : 0;
    const int sgi = (kq >= w0) + (kq >= w0 + w1);
    const int kin = kq - pick3(sgi, 0, w0, w0 + w1);
    const float* base = pick3(sgi, p0, p1, p2) + kin;
    rscp = pick3(sgi, sc0, sc1, sc2) + kin;
    rofp = pick3(sgi, of0, of1, of2) + kin;
    rastr = pick3(sgi, as0, as1, as2);
#pragma unroll
    for (int i = 0; i < AP; ++i) ra[i] = ld4(base + srcoff[sgi * BM + prow + RSTEP * i]);
    if (a.B == 1) {                            // one batch element: one (scale, offset) per chunk
      rsc = ld4(rscp);
      rof = ld4(rofp);
    }
  };
  auto stage_chunk = [&](int c) __attribute__((always_inline)) {
    float* ab = region + (c & 1) * (BM * LDA);
    if constexpr (A16) {
      if (chunk_is_h16(c)) {                   // halfs as stored: group c8 / 4 of the row, 16 bytes into its hi plane
#pragma unroll
        for (int i = 0; i < AP8; ++i) {
          f32x4 v = ra[i];
          if (!rlive) v = f32x4{0.f, 0.f, 0.f, 0.f};
          st4(ab + (prow8 + RSTEP8 * i) * LDA + (c8 >> 2) * 32 + (c8 & 3) * 4, v);
        }
        return;
      }
    }
    with_flag(a.round16, [&](auto rc) __attribute__((always_inline)) {
      constexpr bool RND = decltype(rc)::value;
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        const int row = prow + RSTEP * i;
        f32x4 v;
        if (a.B == 1) {
          v = ra[i] * rsc + rof;
        } else {                               // per-row batch element: (scale, offset) fetched here (cache hits)
          const int bo = srcb[row] * rastr;
          v = ra[i] * ld4(rscp + bo) + ld4(rofp + bo);
        }
        if (!rlive) v = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (F32) st4(ab + row * LDA + c4 * 4, r16_c<RND>(v));
        else stage16<A16>(ab + row * LDA + (c4 >> 3) * 32, c4 & 7, r16_c<RND>(v));
      }
    });
  };

  // ---------------- phase 1: hidden^T = swish(W1^T x concat(segments)^T + b1) ----------------
  const int steps1 = kpad / 16;
  const float* wf1 = a.w1f + (size_t)(wave * NT1) * steps1 * 512 + lane * 4;
  const size_t cts1 = (size_t)steps1 * 512;
  {
    f32x16 acc[MT][NT1], accx[MT][NT1];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          acc[mt][nt][q] = 0.f;
          accx[mt][nt][q] = 0.f;
        }
    f32x4 wh[R1][NT1], wl[R1][NT1];
    ws_ring_fill<NT1, R1>(wh, wl, wf1, cts1, steps1);
    load_chunk(0);
    const int nchunks = (kpad + KC - 1) / KC;
    int s = 0;
    // one chunk: stage, barrier, start the next gather, then the chunk's quads of k16 steps.
    // Buffer c&1 was last read two chunks ago and every wave has passed a barrier since, so one
    // barrier per chunk orders both the refill and the reads.
    auto chunk = [&](int c, auto phase) __attribute__((always_inline)) {
      constexpr int PH = decltype(phase)::value;       // ring slot of the chunk's first step
      stage_chunk(c);
      __syncthreads();
      if (c + 1 < nchunks) load_chunk(c + 1);
      const float* arow = region + (c & 1) * (BM * LDA) + (wrow + r) * LDA + hh * (F32 ? 8 : 4);
      ws_quad<MT, NT1, R1, PH, A16, F32>(acc, accx, wh, wl, arow, 32 * LDA, 0, wf1, cts1, s, steps1);
      if constexpr (KC == 128) {
        if (kpad - c * KC > 64)
          ws_quad<MT, NT1, R1, (PH + 4) % R1, A16, F32>(acc, accx, wh, wl, arow, 32 * LDA, 4, wf1, cts1, s, steps1);
      }
    };
    if constexpr (KC == 128 || R1 == 4) {
      // 8-step chunks (or a 4-step ring): every full chunk starts at ring slot 0; only the last
      // chunk can be a 4-step one, and nothing follows it
      for (int c = 0; c < nchunks; ++c) chunk(c, std::integral_constant<int, 0>{});
    } else {
      // 4-step chunks on an 8-step ring: the chunks alternate between the ring's halves
      int c = 0;
      for (; c + 1 < nchunks; c += 2) {
        chunk(c, std::integral_constant<int, 0>{});
        chunk(c + 1, std::integral_constant<int, 4>{});
      }
      if (c < nchunks) chunk(c, std::integral_constant<int, 0>{});
    }
    __syncthreads();                           // all waves are done with the A chunks: region becomes the hidden tile
    // gathered per-node terms of a split edge MLP: this lane's rows (wrow + mt * 32 + r) of the two add arrays
    const float* ad0[MT];
    const float* ad1[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      ad0[mt] = a.nadd > 0 ? a.add[0].ptr + addoff[wrow + mt * 32 + r] : nullptr;
      ad1[mt] = a.nadd > 1 ? a.add[1].ptr + addoff[BM + wrow + mt * 32 + r] : nullptr;
    }
    with_flag(a.round16, [&](auto rc) __attribute__((always_inline)) {
      constexpr bool RND = decltype(rc)::value;
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt) {
        const int cbase = (wave * NT1 + nt) * 32 + 4 * hh;   // lane's columns: cbase + 8 j + (0..3)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 bv = ld4(a.b1 + cbase + 8 * j);
          f32x4 tadd[MT];                       // (kept local: the accumulators are only READ in here)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) tadd[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (a.nadd > 0) {                     // uniform branch; the row tiles' gathers are in flight together
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              tadd[mt] = ld4(ad0[mt] + cbase + 8 * j);
              if (a.nadd > 1) tadd[mt] += ld4(ad1[mt] + cbase + 8 * j);
            }
          }
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
              v[e] = r16_c<RND>(swish(hilo(acc[mt][nt][4 * j + e], accx[mt][nt][4 * j + e]) + bv[e] + tadd[mt][e]));
            if constexpr (A16)                  // the hidden tile's lo plane is zero and never read
              stage16<true>(region + (size_t)(wrow + mt * 32 + r) * LDH + ((cbase + 8 * j) & ~31), ((cbase + 8 * j) & 31) >> 2,
                            f32x4{v[0], v[1], v[2], v[3]});
            else if constexpr (F32)
              st4(region + (size_t)(wrow + mt * 32 + r) * LDH + cbase + 8 * j, f32x4{v[0], v[1], v[2], v[3]});
            else
              store4_s16(region, (size_t)(wrow + mt * 32 + r), LDH, cbase + 8 * j, v[0], v[1], v[2], v[3]);
          }
        }
      }
    });
  }
  __syncthreads();

  // ---------------- phase 2: y^T = W2^T x hidden^T (+ b2) -------------------------------------
  {
    constexpr int steps2 = HID / 16;
    const float* wf2 = a.w2f + (size_t)(wave * NT2) * steps2 * 512 + lane * 4;
    const size_t cts2 = (size_t)steps2 * 512;
    f32x16 acc[MT][NT2], accx[MT][NT2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          acc[mt][nt][q] = 0.f;
          accx[mt][nt][q] = 0.f;
        }
    f32x4 wh[R2][NT2], wl[R2][NT2];
    const bool p2 = wave < NW2;                   // waves beyond the output width only join the barriers
    if (p2) ws_ring_fill<NT2, R2>(wh, wl, wf2, cts2, steps2);
    int s = 0;
    const float* hrow = region + (wrow + r) * LDH + hh * (F32 ? 8 : 4);
#pragma unroll 1
    for (int st = 0; st < (p2 ? steps2 : 0); st += 8) {       // HID % 128 == 0: steps2 % 8 == 0
      ws_quad<MT, NT2, R2, 0, A16, F32>(acc, accx, wh, wl, hrow, 32 * LDH, st, wf2, cts2, s, steps2);
      ws_quad<MT, NT2, R2, 4 % R2, A16, F32>(acc, accx, wh, wl, hrow, 32 * LDH, st + 4, wf2, cts2, s, steps2);
    }
    __syncthreads();                           // hidden tile no longer needed: region becomes the output tile
    with_flag(a.round16, [&](auto rc) __attribute__((always_inline)) {
      constexpr bool RND = decltype(rc)::value;
#pragma unroll
      for (int nt = 0; nt < (p2 ? NT2 : 0); ++nt) {
        const int cbase = (wave * NT2 + nt) * 32 + 4 * hh;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 bv = ld4(a.b2 + cbase + 8 * j);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = hilo(acc[mt][nt][4 * j + e], accx[mt][nt][4 * j + e]) + bv[e];
            st4(region + (wrow + mt * 32 + r) * LDY + cbase + 8 * j, r16_c<RND>(v));
          }
        }
      }
    });
  }
  // ---------------- row epilogue: LayerNorm, conditioning, residual ----------------------------
  // Each wave finishes its 8*MT rows in ONE pass; a lane owns 4 consecutive columns per 256-column
  // group, so a row costs one 16-byte LDS read, residual load and store (the scalar-column form
  // with 8-row batches took a quarter of the big edge MLP's time: 32 dword loads + 32 dword stores
  // per batch and a dependent conditioning fetch per row).
  const float* Ybuf = region;
  const int n = a.n_out;
  const float inv_n = 1.0f / (float)n;
  constexpr int CG = (NPAD + 255) / 256;       // 256-column groups
  constexpr int RW = 32 * MT / NWC;            // rows per wave
  const int rbase = wave_all * RW;
  const bool vec_io = (a.ldo % 4 == 0) && (n % 4 == 0) &&
                      (!a.cond || ((reinterpret_cast<size_t>(a.cond) & 15) == 0 && a.cond_stride % 4 == 0));
  // The residual rows and (one batch element) the conditioning vectors are requested BEFORE the barrier that
  // publishes the output tile: they depend on nothing computed here, and fetched after the barrier / after the
  // LayerNorm statistics they were two more dependent round trips at the end of every workgroup.
  f32x4 yv[RW][CG], rv[RW][CG], scpre[CG], ofpre[CG];
#pragma unroll
  for (int j = 0; j < CG; ++j) {
    const int c = 4 * lane + 256 * j;
    scpre[j] = f32x4{1.f, 1.f, 1.f, 1.f};
    ofpre[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < n && a.cond && a.B == 1 && vec_io) {       // one conditioning vector for every row
      scpre[j] = ld4(a.cond + c);
      ofpre[j] = ld4(a.cond + n + c);
    }
  }
  if (tri) {
    // ---- triple epilogue: out[u] = f(row 3j) + f(row 3j+1) + f(row 3j+2), f = conditioning(LayerNorm(.)) ----
    // = jraph.segment_sum over the 3 mesh2grid edges of a grid node (deep_typed_graph_net.py:396-410;
    // common/grid_mesh_connectivity.py:118-131), added in ascending edge order like gc_segsum_small_kernel, float32;
    // the updated edges themselves are never stored (the reference discards them: gencast/denoiser.py:765-768).
    // Wave w takes triples w, w + NWV, ...  Host: vec_io holds, no residual.
    constexpr int NWV = NWC * WM, TPW = (TPB + NWV - 1) / NWV;
    __syncthreads();
    with_flag(a.round_out, [&](auto rc) __attribute__((always_inline)) {
      constexpr bool RND = decltype(rc)::value;
#pragma unroll
      for (int k = 0; k < TPW; ++k) {
        const int jt = wave_all + NWV * k, u = unit0 + jt;
        if (jt >= TPB || u >= units) break;            // wave-uniform
        f32x4 y3[3][CG];
#pragma unroll
        for (int sl = 0; sl < 3; ++sl)
#pragma unroll
          for (int j = 0; j < CG; ++j) {
            const int c = 4 * lane + 256 * j;
            f32x4 y = {0.f, 0.f, 0.f, 0.f};
            if (c < NPAD) {
              y = ld4(Ybuf + (3 * jt + sl) * LDY + c);
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (c + e >= n) y[e] = 0.f;
            }
            y3[sl][j] = y;
          }
        float mean3[3] = {0.f, 0.f, 0.f}, rstd3[3] = {1.f, 1.f, 1.f};
        if (a.do_ln) {
          float s1[3], s2[3];
#pragma unroll
          for (int sl = 0; sl < 3; ++sl) {
            s1[sl] = 0.f;
            s2[sl] = 0.f;
#pragma unroll
            for (int j = 0; j < CG; ++j) ln_accumulate(y3[sl][j], s1[sl], s2[sl]);
          }
#pragma unroll
          for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int sl = 0; sl < 3; ++sl) {
              s1[sl] += __shfl_xor(s1[sl], o);
              s2[sl] += __shfl_xor(s2[sl], o);
            }
#pragma unroll
          for (int sl = 0; sl < 3; ++sl) ln_finish(s1[sl], s2[sl], inv_n, mean3[sl], rstd3[sl]);
        }
#pragma unroll
        for (int j = 0; j < CG; ++j) {
          const int c = 4 * lane + 256 * j;
          if (c >= n) continue;
          f32x4 sc = scpre[j], of = ofpre[j];
          if (a.cond && a.B != 1) {
            const float* cs = a.cond + (size_t)(u % a.B) * a.cond_stride;
            sc = ld4(cs + c);
            of = ld4(cs + n + c);
          }
          f32x4 t = r16_c<RND>(ln_affine(y3[0][j], mean3[0], rstd3[0], sc, of));
          t += r16_c<RND>(ln_affine(y3[1][j], mean3[1], rstd3[1], sc, of));
          t += r16_c<RND>(ln_affine(y3[2][j], mean3[2], rstd3[2], sc, of));
          t = r16_c<RND>(t);
          if (A16 && !a.out_f32) sth4(as_h16(a.out) + (size_t)u * a.ldo + c, t);
          else st4(a.out + (size_t)u * a.ldo + c, t);
        }
      }
    });
    return;
  }
#pragma unroll
  for (int j = 0; j < CG; ++j) {
    const int c = 4 * lane + 256 * j;
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
      const int orow = row0 + rbase + rr;
      f32x4 rs = {0.f, 0.f, 0.f, 0.f};
      if (a.residual && orow < a.rows && c < n) {
        if constexpr (A16) {                   // the residual is a stored activation: halfs
          if (vec_io) {
            rs = ldh4(as_h16(a.residual) + (size_t)orow * n + c);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (c + e < n) rs[e] = (float)as_h16(a.residual)[(size_t)orow * n + c + e];
          }
        } else if (vec_io) {
          rs = ld4(a.residual + (size_t)orow * n + c);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c + e < n) rs[e] = a.residual[(size_t)orow * n + c + e];
        }
      }
      rv[rr][j] = rs;
    }
  }
  __syncthreads();
#pragma unroll
  for (int rr = 0; rr < RW; ++rr) {
#pragma unroll
    for (int j = 0; j < CG; ++j) {
      const int c = 4 * lane + 256 * j;
      f32x4 y = {0.f, 0.f, 0.f, 0.f};
      if (c < NPAD) {
        y = ld4(Ybuf + (rbase + rr) * LDY + c);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c + e >= n) y[e] = 0.f;          // padded output columns stay out of the statistics
      }
      yv[rr][j] = y;
    }
  }
  float mean[RW], rstd[RW];
  if (a.do_ln) {
    float s1[RW], s2[RW];
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
      s1[rr] = 0.f;
      s2[rr] = 0.f;
#pragma unroll
      for (int j = 0; j < CG; ++j) ln_accumulate(yv[rr][j], s1[rr], s2[rr]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int rr = 0; rr < RW; ++rr) {
        s1[rr] += __shfl_xor(s1[rr], o);
        s2[rr] += __shfl_xor(s2[rr], o);
      }
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) ln_finish(s1[rr], s2[rr], inv_n, mean[rr], rstd[rr]);
  } else {
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
      mean[rr] = 0.f;
      rstd[rr] = 1.f;
    }
  }
  with_flag(a.round_out, [&](auto rc) __attribute__((always_inline)) {
  constexpr bool RND = decltype(rc)::value;
#pragma unroll
  for (int j = 0; j < CG; ++j) {
    const int c = 4 * lane + 256 * j;
    if (c >= n) continue;
    const f32x4 sc1 = scpre[j], of1 = ofpre[j];
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
      const int orow = row0 + rbase + rr;
      if (orow >= a.rows) break;
      f32x4 sc = sc1, of = of1;
      if (a.cond && !(a.B == 1 && vec_io)) {
        const float* cs = a.cond + (size_t)(orow % a.B) * a.cond_stride;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c + e < n) {
            sc[e] = cs[c + e];
            of[e] = cs[n + c + e];
          }
      }
      const f32x4 v = r16_c<RND>(r16_c<RND>(ln_affine(yv[rr][j], mean[rr], rstd[rr], sc, of)) + rv[rr][j]);
      if (A16 && !a.out_f32) {                 // the output is a stored activation: halfs (v is fp16-exact: RND)
        if (vec_io) {
          sth4(as_h16(a.out) + (size_t)orow * a.ldo + c, v);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c + e < n) as_h16(a.out)[(size_t)orow * a.ldo + c + e] = (_Float16)v[e];
        }
      } else if (vec_io) {
        if (a.wt) st4_wt(a.out + (size_t)orow * a.ldo + c, v);
        else st4(a.out + (size_t)orow * a.ldo + c, v);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c + e < n) a.out[(size_t)orow * a.ldo + c + e] = v[e];
      }
    }
  }
  });
}

template <int NT1, int NT2, int MT, int WM, int NWC = 4, int NW2 = NWC,
          int OCC = mlp_ws_occ<NT1, WM, NWC>() /* waves per SIMD the registers are held to */, bool A16 = false,
          bool F32 = false>
__global__ __launch_bounds__(64 * NWC * WM, OCC) void gc_mlp_ws_kernel(MlpArgs a) {
  gc_mlp_ws_body<NT1, NT2, MT, WM, NWC, NW2, OCC, A16, F32>(a, (int)blockIdx.x);
}

// Two independent fused MLPs of one shape in one launch (the grid2mesh edge update and the grid-node update both read
// only g0 / m0: typed_graph_net.py:134-195).  At nano the edge update's 526 tiles are one round of 512 workgroups plus a
// round of 14 that costs as much again; the node update's 329 tiles fill that second round instead of a launch of their own.
template <int NT1, int NT2, int MT, int WM, int NWC = 4, int NW2 = NWC,
          int OCC = mlp_ws_occ<NT1, WM, NWC>(), bool A16 = false, bool F32 = false>
__global__ __launch_bounds__(64 * NWC * WM, OCC) void gc_mlp_ws_pair_kernel(MlpArgs a, MlpArgs b, int na) {
  if ((int)blockIdx.x < na) gc_mlp_ws_body<NT1, NT2, MT, WM, NWC, NW2, OCC, A16, F32>(a, (int)blockIdx.x);
  else gc_mlp_ws_body<NT1, NT2, MT, WM, NWC, NW2, OCC, A16, F32>(b, (int)blockIdx.x - na);
}

// argument checks + launch geometry shared by the single and the pair launcher
template <int NT1, int NT2, int MT, int WM, int NWC, int NW2>
static hipError_t mlp_ws_geometry(const MlpArgs& a, int* grid_out, size_t* lds_out) {
  constexpr int HID = NT1 * 32 * NWC, NPAD = NT2 * 32 * NW2, BM = 32 * MT * WM;
  constexpr int abuf = 2 * BM * ((BM >= 128 ? 64 : 128) + 4);
  constexpr int region = (abuf > BM * (HID + 4)) ? abuf : BM * (HID + 4);
  static_assert(BM * (NPAD + 4) <= region, "output tile must fit the shared region");
  *lds_out = (size_t)(6 * BM + region) * sizeof(float);
  if (a.nadd < 0 || a.nadd > 2 || (a.nadd > 0 && !a.add[0].ptr) || (a.nadd > 1 && !a.add[1].ptr))
    return hipErrorInvalidValue;
  int ksum = 0;
  for (int i = 0; i < a.nseg; ++i) {
    if (a.seg[i].width % 32 || a.seg[i].ld % 4) return hipErrorInvalidValue;
    ksum += a.seg[i].width;
  }
  if (a.k1f % 64 || a.k1f < ksum || a.k1f - ksum >= 64 || !a.w2f || !a.ones || !a.zeros || ksum > 512 * 3)
    return hipErrorInvalidValue;
  if (kTuA16) {                                 // fp16-stored segments: 16-byte pieces of 8 halfs, one segment per K chunk
    constexpr int KCc = BM >= 128 ? 64 : 128;
    for (int i = 0; i < a.nseg; ++i) {
      const bool h16 = !(i == 0 && a.seg0_f32);
      if (h16 && (a.seg[i].affine || a.seg[i].ld % 8 || a.seg[i].width % 8)) return hipErrorInvalidValue;
      if (a.nseg > 1 && a.seg[i].width % KCc) return hipErrorInvalidValue;
    }
  }
  int grid = (a.rows + BM - 1) / BM;
  if (a.tri) {   // rows = 3 x units; a tile holds BM / 3 whole triples; the epilogue's 16-byte column ownership is required
    if (a.rows % 3 || a.residual || a.ldo % 4 || a.n_out % 4 ||
        (a.cond && ((reinterpret_cast<size_t>(a.cond) & 15) || a.cond_stride % 4)))
      return hipErrorInvalidValue;
    grid = (a.rows / 3 + BM / 3 - 1) / (BM / 3);
  }
  *grid_out = grid;
  return hipSuccess;
}

template <int NT1, int NT2, int MT, int WM, int NWC = 4, int NW2 = NWC>
static hipError_t launch_mlp_ws_pair_t(hipStream_t s, const MlpArgs& a, const MlpArgs& b) {
  int ga = 0, gb = 0;
  size_t lds = 0;
  if (hipError_t e = mlp_ws_geometry<NT1, NT2, MT, WM, NWC, NW2>(a, &ga, &lds)) return e;
  if (hipError_t e = mlp_ws_geometry<NT1, NT2, MT, WM, NWC, NW2>(b, &gb, &lds)) return e;
  constexpr int OCC = mlp_ws_occ<NT1, WM, NWC>();
  if constexpr (!kTuA16) {
    if (a.f32w) {
      static DynLdsOnce once32;
      if (hipError_t e = once32.ensure((const void*)gc_mlp_ws_pair_kernel<NT1, NT2, MT, WM, NWC, NW2, OCC, false, true>, (int)lds)) return e;
      hipLaunchKernelGGL((gc_mlp_ws_pair_kernel<NT1, NT2, MT, WM, NWC, NW2, OCC, false, true>), dim3(ga + gb),
                         dim3(64 * NWC * WM), lds, s, a, b, ga);
      return hipGetLastError();
    }
  }
  static DynLdsOnce once;
  if (hipError_t e = once.ensure((const void*)gc_mlp_ws_pair_kernel<NT1, NT2, MT, WM, NWC, NW2, OCC, kTuA16>, (int)lds)) return e;
  hipLaunchKernelGGL((gc_mlp_ws_pair_kernel<NT1, NT2, MT, WM, NWC, NW2, OCC, kTuA16>), dim3(ga + gb),
                     dim3(64 * NWC * WM), lds, s, a, b, ga);
  return hipGetLastError();
}

template <int NT1, int NT2, int MT, int WM, int NWC = 4, int NW2 = NWC>
static hipError_t launch_mlp_ws_t(hipStream_t s, const MlpArgs& a) {
  int grid = 0;
  size_t lds = 0;
  if (hipError_t e = mlp_ws_geometry<NT1, NT2, MT, WM, NWC, NW2>(a, &grid, &lds)) return e;
  constexpr int OCC = mlp_ws_occ<NT1, WM, NWC>();
  if constexpr (!kTuA16) {
    if (a.f32w) {
      static DynLdsOnce once32;
      if (hipError_t e = once32.ensure((const void*)gc_mlp_ws_kernel<NT1, NT2, MT, WM, NWC, NW2, OCC, false, true>, (int)lds)) return e;
      hipLaunchKernelGGL((gc_mlp_ws_kernel<NT1, NT2, MT, WM, NWC, NW2, OCC, false, true>), dim3(grid),
                         dim3(64 * NWC * WM), lds, s, a);
      return hipGetLastError();
    }
  }
  static DynLdsOnce once;
  if (hipError_t e = once.ensure((const void*)gc_mlp_ws_kernel<NT1, NT2, MT, WM, NWC, NW2, OCC, kTuA16>, (int)lds)) return e;
  hipLaunchKernelGGL((gc_mlp_ws_kernel<NT1, NT2, MT, WM, NWC, NW2, OCC, kTuA16>), dim3(grid),
                     dim3(64 * NWC * WM), lds, s, a);
  return hipGetLastError();
}

template <int NT1, int NT2, bool F16, int WM>
static hipError_t launch_mlp_t(hipStream_t s, const MlpArgs& a) {
  const int hidden = NT1 * 128, n_pad = NT2 * 128, bm = 32 * WM;
  const int wrows = hidden > n_pad ? hidden : n_pad;
  const size_t lds = (size_t)(wrows * kMlpLd + bm * kMlpLd + bm * (hidden + 4)) * sizeof(float);
  for (int i = 0; i < a.nseg; ++i)
    if (a.seg[i].width % kMlpBK || a.seg[i].ld % 4) return hipErrorInvalidValue;
  static DynLdsOnce once;
  if (hipError_t e = once.ensure((const void*)gc_mlp_kernel<NT1, NT2, F16, WM>, (int)lds)) return e;
  const int grid = (a.rows + bm - 1) / bm;
  hipLaunchKernelGGL((gc_mlp_kernel<NT1, NT2, F16, WM>), dim3(grid), dim3(256 * WM), lds, s, a);
  return hipGetLastError();
}

bool mlp_runs_weight_streaming(const MlpArgs& a) {
  const int nt1 = a.hidden / 128, nt2 = a.n_out_pad / 128;
  if (a.hidden % 128 || a.n_out_pad % 128 || a.n_out_pad > a.hidden || !((a.f16 || a.f32w) && a.w1f)) return false;
  if (nt1 <= 2) return nt2 <= nt1;
  const char* e = getenv("GC_TUNE_MLP_WS512");
  return nt1 == 4 && !(e && *e && atoi(e) == 0) && (nt2 == 4 || nt2 == 1);
}

hipError_t launch_mlp(hipStream_t s, const MlpArgs& a) {
  const int nt1 = a.hidden / 128, nt2 = a.n_out_pad / 128;
  if (a.hidden % 128 || a.n_out_pad % 128 || a.n_out_pad > a.hidden) return hipErrorInvalidValue;
  // 64-row tiles once there are enough rows to keep every CU busy with them (f16x3 only: the f32
  // path is MFMA-bound and prefers more, smaller tiles)
  static int big_rows = -1;
  if (big_rows < 0) {
    const char* e = getenv("GC_TUNE_MLP_BIG_ROWS");
    big_rows = (e && *e) ? atoi(e) : (1 << 30);   // measured neutral at nano: off by default
  }
  // weight-streaming form; hidden = 512 runs it with 8 column waves (GC_TUNE_MLP_WS512=0: LDS-staged kernel)
  static int ws512 = -1;
  if (ws512 < 0) {
    const char* e = getenv("GC_TUNE_MLP_WS512");
    ws512 = (e && *e) ? atoi(e) : 2;             // != 0: 64-row tiles (1-degree config: 3.1 ms per call; LDS-staged 7.1)
  }
  const bool ws_ok = (a.f16 || a.f32w) && a.w1f;   // weight-streaming form: f16x3 (WF16 images) or exact f32 (WF32)
  if (ws_ok && nt1 == 4 && ws512) {
    if (nt2 == 4) return launch_mlp_ws_t<2, 2, 2, 1, 8, 8>(s, a);   // (the 32-row forms: 3.9 vs 3.1 ms per 1-degree call, removed in round 5)
    if (nt2 == 1) return launch_mlp_ws_t<2, 1, 2, 1, 8, 4>(s, a);
  }
  // hidden = 256 below the 64-row threshold: 8 column waves x 32 rows, two workgroups (16 waves) per
  // CU -- measured 1-6 us faster per launch than 4 waves x 32 rows (GC_TUNE_MLP_WS8=0 for the latter)
  static int ws8 = -1, ws8_rows = -1;
  if (ws8 < 0) {
    const char* e = getenv("GC_TUNE_MLP_WS8");
    ws8 = (e && *e) ? atoi(e) : 1;
    const char* r = getenv("GC_TUNE_MLP_MT2_ROWS");
    ws8_rows = (r && *r) ? atoi(r) : 24000;
  }
  if (ws_ok && nt1 == 2 && ws8 && a.rows < ws8_rows) {
    if (nt2 == 2) return launch_mlp_ws_t<1, 1, 1, 1, 8, 8>(s, a);
    if (nt2 == 1) return launch_mlp_ws_t<1, 1, 1, 1, 8, 4>(s, a);
  }
  if (ws_ok && nt1 <= 2) {
    static int mt2_rows = -1;                   // 64-row tiles from this many rows on (GC_TUNE_MLP_MT2_ROWS)
    if (mt2_rows < 0) {
      const char* e = getenv("GC_TUNE_MLP_MT2_ROWS");
      mt2_rows = (e && *e) ? atoi(e) : 24000;   // measured at nano: 64-row tiles win only on the 31.5k-edge MLP
    }
    // (128-row tiles with one workgroup per CU were measured slower than 64-row ones -- 84 vs 77 us on the nano
    //  mesh2grid edge MLP -- and left the build in round 5 together with the NT1 = 4 four-wave forms no dispatch reached)
    const bool mt2 = a.rows >= mt2_rows;
#define GC_MLP_WS(A_, B_)                                                                \
  if (nt1 == A_ && nt2 == B_) {                                                          \
    if (mt2) return launch_mlp_ws_t<A_, B_, 2, 1>(s, a);                                 \
    return launch_mlp_ws_t<A_, B_, 1, 1>(s, a);                                          \
  }
    GC_MLP_WS(1, 1) GC_MLP_WS(2, 2) GC_MLP_WS(2, 1)
#undef GC_MLP_WS
    return hipErrorInvalidValue;
  }
  // from here on: the LDS-staged kernels, which read and write float32 arrays only.  With physical fp16 storage
  // (a.a16: halfs behind the float* fields) that would be silently wrong values -- e.g. GC_TUNE_MLP_WS512=0 at hidden
  // 512 while gc_api's store16_ok() looks at the other switches only -- so it is an error instead
  if (a.a16 || a.tri) return hipErrorInvalidValue;   // (the triple epilogue exists in the weight-streaming kernel only)
  const bool big = a.f16 && a.rows >= big_rows && nt1 <= 2;
#define GC_MLP(A_, B_)                                                                   \
  if (nt1 == A_ && nt2 == B_) {                                                          \
    if (!a.f16) return launch_mlp_t<A_, B_, false, 1>(s, a);                             \
    if constexpr (A_ <= 2) { if (big) return launch_mlp_t<A_, B_, true, 2>(s, a); }      \
    return launch_mlp_t<A_, B_, true, 1>(s, a);                                          \
  }
  GC_MLP(1, 1) GC_MLP(2, 2) GC_MLP(2, 1) GC_MLP(4, 4) GC_MLP(4, 1)
#undef GC_MLP
  return hipErrorInvalidValue;
}

// Both MLPs on the 8-wave x 32-row weight-streaming form of hidden 256 (what launch_mlp picks below 24 000 rows), same
// precision family, no triple epilogue: then one launch can run them side by side.
bool mlp_pair_supported(const MlpArgs& a, const MlpArgs& b) {
  auto ok = [](const MlpArgs& m) {
    if (!((m.f16 || m.f32w) && m.w1f) || m.tri) return false;
    if (m.hidden == 512 && m.n_out_pad == 512) {          // the 8-wave x 64-row form of latent 512
      const char* e = getenv("GC_TUNE_MLP_WS512");
      return !(e && *e && atoi(e) == 0);
    }
    const char* e8 = getenv("GC_TUNE_MLP_WS8");
    const char* r8 = getenv("GC_TUNE_MLP_MT2_ROWS");
    const int ws8_rows = (r8 && *r8) ? atoi(r8) : 24000;
    return m.hidden == 256 && m.n_out_pad == 256 && !(e8 && *e8 && atoi(e8) == 0) && m.rows < ws8_rows;
  };
  return ok(a) && ok(b) && a.hidden == b.hidden && a.f16 == b.f16 && a.f32w == b.f32w && a.a16 == b.a16;
}
hipError_t launch_mlp_pair(hipStream_t s, const MlpArgs& a, const MlpArgs& b) {
  if (!mlp_pair_supported(a, b)) return hipErrorInvalidValue;
  if (a.hidden == 512) return launch_mlp_ws_pair_t<2, 2, 2, 1, 8, 8>(s, a, b);
  return launch_mlp_ws_pair_t<1, 1, 1, 1, 8, 8>(s, a, b);
}

// ----------------------------------------------------------------------------
// gc_segsum: jraph.segment_sum by receiver (deep_typed_graph_net.py:396-410;
// typed_graph_net.py:175-182) as a CSR walk: one wave per output row, edges
// added in ascending edge id -> deterministic, no atomics.
// ----------------------------------------------------------------------------
// H16: src and out are _Float16 arrays (physical fp16 activation storage); the sums stay float32.
template <bool H16>
__device__ __forceinline__ f32x4 seg_ld4(const float* base, size_t elem) {
  if constexpr (H16) return ldh4(as_h16(base) + elem);
  else return ld4(base + elem);
}
template <bool H16>
__device__ __forceinline__ void seg_st4(float* base, size_t elem, f32x4 v) {
  if constexpr (H16) sth4(as_h16(base) + elem, v);
  else st4(base + elem, v);
}

template <bool H16>
__global__ __launch_bounds__(256) void gc_segsum_kernel(const float* __restrict__ src,
                                                         const int* __restrict__ rowptr,
                                                         const int* __restrict__ eids, int n_items,
                                                         int B, int width, float* __restrict__ out, int round16,
                                                         float norm /* 0: none; else the sum is divided by it (f32) */) {
  // One workgroup per output row: wave w adds edges e0+w, e0+w+4, ... , then the four partial rows are added
  // in wave order through LDS.  The grid2mesh in-degree is very skewed (3 ... 218 at 2.5 deg, more at 1 deg:
  // pole mesh nodes), so a row must not be one wave's job -- and the pole rows ARE the kernel's duration: with
  // `id = eids[e]; s += src[id]` per edge, a wave's 55 edges were 55 x 2 dependent memory round trips.  Now a
  // wave fetches the ids of up to 64 of its edges with one load per lane and keeps 8 row loads in flight.
  // Order of the additions (and so the bits): edge k of the wave goes to accumulator k & 1, ascending k.
  __shared__ __attribute__((aligned(16))) float part[4][512];
  const int wrow = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int item = wrow / B, b = wrow - item * B;
  const int e0 = rowptr[item], e1 = rowptr[item + 1];
  const int n_w = (e1 - e0 - wave + 3) >> 2;                    // edges of this wave (may be <= 0)
  for (int c0 = 0; c0 < width; c0 += 256) {
    const int c = c0 + lane * 4;
    const int cc = c < width ? c : 0;                            // lanes beyond the width read column 0, store nothing
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    for (int kb = 0; kb < n_w; kb += 64) {
      const int mine = e0 + wave + 4 * (kb + lane);
      const int id = eids[mine < e1 ? mine : e1 - 1];
      const int cnt = (n_w - kb) < 64 ? (n_w - kb) : 64;
      for (int j = 0; j < cnt; j += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int ju = (j + u < cnt) ? j + u : cnt - 1;        // uniform: beyond the end, re-read the last row
          const int idj = __builtin_amdgcn_readlane(id, ju);
          v[u] = seg_ld4<H16>(src, ((size_t)idj * B + b) * width + cc);
        }
        __builtin_amdgcn_sched_barrier(0);                       // all eight requested before the first is used
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (j + u < cnt) {
            if (u & 1) s1 += v[u];
            else s0 += v[u];
          }
      }
    }
    s0 += s1;
    if (c < width) st4(&part[wave][lane * 4], s0);
    __syncthreads();
    if (wave == 0 && c < width) {
      f32x4 t = ld4(&part[0][lane * 4]);
      t += ld4(&part[1][lane * 4]);
      t += ld4(&part[2][lane * 4]);
      t += ld4(&part[3][lane * 4]);
      if (norm != 0.f) t = t / norm;            // aggregate_normalization (deep_typed_graph_net.py:396-410): f32, before the cast
      seg_st4<H16>(out, (size_t)wrow * width + c, r16_if(t, round16));
    }
    __syncthreads();
  }
}

// Low, even in-degree (mesh2grid: exactly 3 edges per grid node): one wave per output row; the ids of four
// edges, then their four rows, are requested together (one by one it was rowptr -> id -> row -> id -> row ...:
// seven dependent round trips for three edges).  Added in ascending edge order.
template <bool H16>
__global__ __launch_bounds__(256) void gc_segsum_small_kernel(const float* __restrict__ src,
                                                               const int* __restrict__ rowptr,
                                                               const int* __restrict__ eids, int n_items,
                                                               int B, int width, float* __restrict__ out, int round16,
                                                               float norm) {
  const int wrow = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (wrow >= n_items * B) return;
  const int item = wrow / B, b = wrow - item * B;
  const int e0 = rowptr[item], e1 = rowptr[item + 1];
  for (int c = lane * 4; c < width; c += 256) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int e = e0; e < e1; e += 4) {
      int id[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) id[u] = eids[(e + u < e1) ? e + u : e1 - 1];
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = seg_ld4<H16>(src, ((size_t)id[u] * B + b) * width + c);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (e + u < e1) acc += v[u];
    }
    if (norm != 0.f) acc = acc / norm;
    seg_st4<H16>(out, (size_t)wrow * width + c, r16_if(acc, round16));
  }
}

hipError_t launch_segsum(hipStream_t s, const float* src, const int* rowptr, const int* eids,
                         int n_items, int n_edges, int B, int width, float* out, bool round16, bool h16, float norm) {
  if (width % 4 || width > 512) return hipErrorInvalidValue;
  const dim3 gs((n_items * B + 3) / 4), gl(n_items * B), blk(256);
  if (n_edges <= 4 * n_items) {
    if (h16) hipLaunchKernelGGL(gc_segsum_small_kernel<true>, gs, blk, 0, s, src, rowptr, eids, n_items, B, width, out, 1, norm);
    else hipLaunchKernelGGL(gc_segsum_small_kernel<false>, gs, blk, 0, s, src, rowptr, eids, n_items, B, width, out, round16 ? 1 : 0, norm);
  } else {
    if (h16) hipLaunchKernelGGL(gc_segsum_kernel<true>, gl, blk, 0, s, src, rowptr, eids, n_items, B, width, out, 1, norm);
    else hipLaunchKernelGGL(gc_segsum_kernel<false>, gl, blk, 0, s, src, rowptr, eids, n_items, B, width, out, round16 ? 1 : 0, norm);
  }
  return hipGetLastError();
}

// ----------------------------------------------------------------------------
// gc_gemm: C = A @ W (+bias, +gelu) or split-K partial slabs, on a (32*MT) x 128
// block tile.  Both operands are staged through LDS in 32-wide K tiles by fully
// coalesced 16-byte loads (8 consecutive lanes fetch one 128-byte row segment),
// double-buffered with the next tile's global loads in flight during the MFMAs.
// 4 waves; wave w owns columns [32w, 32w+32) and all 32*MT rows.
// Used for QKV (sparse_transformer.py:271-290), attention out-projection (:351-353)
// and both FFW layers (:252-268).  grid = (row tiles, n/128, k splits).
// ----------------------------------------------------------------------------
constexpr int kBK = 32;
constexpr int kLdT = kBK + 4;  // 36 floats = 9 x 16 B: conflict-free ds_read_b128 / ds_write_b128

// Persistent form: a workgroup walks output tiles t = blockIdx.x, +gridDim.x, ... and treats
// (tile, k-tile) pairs as one stream, so the first operands of the NEXT output tile are already
// in flight while the current tile finishes and its epilogue stores drain: the per-tile
// prologue latency and the store tail overlap with MFMA work instead of adding to it.
// Tile order: t -> (panel = W column tile x k-split, row tile); when the panel count is a
// multiple of 8, blocks with equal (blockIdx % 8) -- the ones that share an XCD's L2 -- work on
// the same W panels (placement is only ever a speed matter, never correctness).
// Workgroup shape: WM x WN waves; each wave owns MT x NT 32x32 accumulator tiles, so the block
// tile is (32*MT*WM) x (32*NT*WN).  Instantiated: 1x4 waves, MT x 1 tiles -> (32*MT) x 128, 256 threads
// (a 4x2-wave 128x128 tile and an LDS-DMA operand ring were measured and dropped: DESIGN.md section 5).
// AMODE 1: the A operand is the attention output, merged on the fly from the S per-key-split
// partials (m, l, unnormalised O) the attention kernel left behind -- the out-projection then
// needs no separate combine launch.
constexpr int kMaxAttnSplits = 8;
template <int WM, int WN, int MT, int NT, int EPI, int CLS, bool F16, int AMODE>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 4 && MT == 1) ? 3 : 2) void gc_gemm_kernel(GemmArgs g) {
  constexpr int BM = 32 * MT * WM;
  constexpr int BN = 32 * NT * WN;
  constexpr int NTHR = 64 * WM * WN;
  constexpr int RPP = NTHR / 8;               // rows staged per pass (8 lanes fetch one 128-B row piece)
  constexpr int AL = BM / RPP, WL = BN / RPP; // 16-byte pieces per thread per K tile
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile must be a multiple of the staging pass");
  __shared__ __attribute__((aligned(16))) float As[2][BM][kLdT];
  __shared__ __attribute__((aligned(16))) float Ws[2][BN][kLdT];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int arow0 = wm * MT * 32, wcol0 = wn * NT * 32;   // this wave's corner inside the block tile
  const int nk = g.k_slice / kBK;
  const int n_mtiles = (g.rows + BM - 1) / BM;
  const int n_panels = (g.n / BN) * g.splits;
  const int total = n_mtiles * n_panels;
  const int lrow = tid >> 3, lc4 = tid & 7;

  auto decode = [&](int t, int& mtile, int& ntile, int& z) {
    int panel;
    if ((n_panels & 7) == 0) {
      const int x = t & 7, q = t >> 3;
      panel = x + 8 * (q / n_mtiles);
      mtile = q % n_mtiles;
    } else {
      panel = t / n_mtiles;
      mtile = t % n_mtiles;
    }
    ntile = panel % (g.n / BN);
    z = panel / (g.n / BN);
  };

  static_assert(AMODE == 0 || AL == 1, "the combining A loader handles one 16-byte piece per thread");
  const float* a_src[AL];
  const float* w_src[WL];
  size_t att_slot0 = 0;                        // AMODE 1: slot of (tile, split 0, batch, head 0)
  int att_q = 0, att_k0 = 0;                   //          query row inside the tile, first column
  auto set_src = [&](int t) {
    int mtile, ntile, z;
    decode(t, mtile, ntile, z);
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      int grow = mtile * BM + lrow + RPP * i;
      if (grow >= g.rows) grow = g.rows - 1;
      a_src[i] = g.a + (size_t)grow * g.lda + z * g.k_slice + lc4 * 4;
      if constexpr (AMODE == 1) {
        const int node = grow / g.att_B, bb = grow - node * g.att_B;
        att_slot0 = ((size_t)(node / kTileM) * g.att_S * g.att_B + bb) * g.att_H;
        att_q = node % kTileM;
        att_k0 = z * g.k_slice + lc4 * 4;
      }
    }
#pragma unroll
    for (int i = 0; i < WL; ++i)
      w_src[i] = g.wt + (size_t)(ntile * BN + lrow + RPP * i) * g.ldw + z * g.k_slice + lc4 * 4;
  };

  // ---- operand stream: positions (tile, kt) in the order this workgroup consumes them ----
  // Loads run TWO positions ahead of the MFMAs through two register sets; LDS is
  // double-buffered one position ahead.
  int t_l = blockIdx.x, kt_l = 0;             // loader position
  if (t_l >= total) return;
  set_src(t_l);
  auto loader_valid = [&]() { return t_l < total; };
  auto loader_advance = [&]() {
    if (++kt_l == nk) {
      kt_l = 0;
      t_l += gridDim.x;
      if (t_l < total) set_src(t_l);
    }
  };
  struct ASet {                                // one in-flight A piece set (plain, or attention partials)
    f32x4 ra[AL];
    f32x4 po[AMODE == 1 ? kMaxAttnSplits : 1];
    float pm[AMODE == 1 ? kMaxAttnSplits : 1], pl[AMODE == 1 ? kMaxAttnSplits : 1];
  };
  ASet ra0, ra1;
  f32x4 rw0[WL], rw1[WL];
  auto load_a = [&](ASet& sa) {
    if constexpr (AMODE == 0) {
#pragma unroll
      for (int i = 0; i < AL; ++i) sa.ra[i] = ld4(a_src[i] + kt_l * kBK);
    } else {
      const int col = att_k0 + kt_l * kBK;             // first of this thread's 4 columns
      const int head = col / g.att_DH, dv = col - head * g.att_DH;
#pragma unroll
      for (int sp = 0; sp < kMaxAttnSplits; ++sp)
        if (sp < g.att_S) {
          const size_t slot = att_slot0 + (size_t)sp * g.att_B * g.att_H + head;
          sa.po[sp] = ld4(g.att_po + slot * (kTileM * g.att_DH) + att_q * g.att_DH + dv);
          sa.pm[sp] = g.att_pml[slot * (kTileM * 2) + att_q * 2];
          sa.pl[sp] = g.att_pml[slot * (kTileM * 2) + att_q * 2 + 1];
        }
    }
  };
  auto merged_a = [&](const ASet& sa) -> f32x4 {       // O = sum_s e^{m_s-m*} O_s / sum_s e^{m_s-m*} l_s
    float mstar = -1e30f;
#pragma unroll
    for (int sp = 0; sp < kMaxAttnSplits; ++sp)
      if (sp < g.att_S) mstar = fmaxf(mstar, sa.pm[sp]);
    f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
    float lsum = 0.f;
#pragma unroll
    for (int sp = 0; sp < kMaxAttnSplits; ++sp)
      if (sp < g.att_S) {
        const float w = (sa.pl[sp] != 0.f) ? __expf(sa.pm[sp] - mstar) : 0.f;
        acc4 += sa.po[sp] * w;
        lsum += w * sa.pl[sp];
      }
    return r16_if(acc4 * ((lsum != 0.f) ? 1.0f / lsum : 0.f), g.round16);
  };
#define GC_LOAD(RA, RW)                                                       \
  {                                                                           \
    load_a(RA);                                                               \
    _Pragma("unroll") for (int i = 0; i < WL; ++i) RW[i] = ld4(w_src[i] + kt_l * kBK); \
  }
#define GC_STAGE(RA, RW, B)                                                   \
  {                                                                           \
    _Pragma("unroll") for (int i = 0; i < AL; ++i) {                          \
      const f32x4 av = (AMODE == 1) ? merged_a(RA) : RA.ra[i];                \
      if (F16 && (g.a_f32 || AMODE == 1)) stage_split16(&As[B][lrow + RPP * i][0], lc4, av); \
      else st4(&As[B][lrow + RPP * i][lc4 * 4], av);                          \
    }                                                                         \
    _Pragma("unroll") for (int i = 0; i < WL; ++i) st4(&Ws[B][lrow + RPP * i][lc4 * 4], RW[i]); \
  }
  GC_LOAD(ra0, rw0);
  loader_advance();
  bool have_next = loader_valid();   // does the "next" register set hold position c+1?
  if (have_next) {
    GC_LOAD(ra1, rw1);
    loader_advance();
  }
  GC_STAGE(ra0, rw0, 0);
  __syncthreads();

  f32x16 acc[MT][NT], acc2[MT][NT];            // acc2: the 2048-scaled cross terms of the f16x3 form
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        acc[mt][nt][q] = 0.f;
        acc2[mt][nt][q] = 0.f;
      }

  int t = blockIdx.x, kt = 0;                 // compute position
  // bias of the current output tile, fetched when the tile starts (a load in the epilogue would
  // stall every wave for an L2 round trip per tile)
  float bias_reg[NT];
  auto fetch_bias = [&](int tt) {
    int mtile, ntile, z;
    decode(tt, mtile, ntile, z);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      bias_reg[nt] = (EPI != 1 && g.bias) ? g.bias[ntile * BN + wcol0 + nt * 32 + r] : 0.f;
  };
  fetch_bias(t);
  // One step: (1) refill the register set that was staged last step with position c+2,
  // (2) MFMAs of position c from LDS[B], (3) stage position c+1 (other register set) into
  // LDS[B^1], (4) barrier, (5) epilogue when position c closed an output tile.
#define GC_STEP(RA_FREE, RW_FREE, RA_NEXT, RW_NEXT, HAVE_NEXT, B)             \
  {                                                                           \
    const bool refill = loader_valid();                                       \
    if (refill) {                                                             \
      GC_LOAD(RA_FREE, RW_FREE);                                              \
      loader_advance();                                                       \
    }                                                                         \
    if constexpr (!F16) {                                                     \
      f32x4 b4[NT][4];                                                        \
      _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                       \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) b4[nt][q] = ld4(&Ws[B][wcol0 + nt * 32 + r][hh * 16 + 4 * q]); \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                     \
        f32x4 a4[4];                                                          \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) a4[q] = ld4(&As[B][arow0 + mt * 32 + r][hh * 16 + 4 * q]); \
        _Pragma("unroll") for (int q = 0; q < 4; ++q)                         \
          _Pragma("unroll") for (int e = 0; e < 4; ++e)                       \
            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                 \
              acc[mt][nt] = mfma32(a4[q][e], b4[nt][q][e], acc[mt][nt]);      \
      }                                                                       \
    } else {                                                                  \
      /* S16 tile row = [hi: 32 halfs | lo: 32 halfs]; k-step ks, lane half hh -> halfs 16ks+8hh.. */ \
      f32x4 bh[NT][2], bl[NT][2];                                             \
      _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                       \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                    \
          bh[nt][ks] = ld4(&Ws[B][wcol0 + nt * 32 + r][ks * 8 + hh * 4]);     \
          bl[nt][ks] = ld4(&Ws[B][wcol0 + nt * 32 + r][16 + ks * 8 + hh * 4]); \
        }                                                                     \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                       \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                    \
          const f32x4 ah = ld4(&As[B][arow0 + mt * 32 + r][ks * 8 + hh * 4]); \
          const f32x4 al = ld4(&As[B][arow0 + mt * 32 + r][16 + ks * 8 + hh * 4]); \
          _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                 \
            acc[mt][nt] = mfma16(ah, bh[nt][ks], acc[mt][nt]);                \
            acc2[mt][nt] = mfma16(ah, bl[nt][ks], acc2[mt][nt]);              \
            acc2[mt][nt] = mfma16(al, bh[nt][ks], acc2[mt][nt]);              \
          }                                                                   \
        }                                                                     \
    }                                                                         \
    if (HAVE_NEXT) GC_STAGE(RA_NEXT, RW_NEXT, (B) ^ 1);                       \
    __syncthreads();                                                          \
    HAVE_NEXT = refill;   /* the set just refilled is "next" two steps from now */ \
    if (++kt == nk) {                                                         \
      epilogue(t);                                                            \
      kt = 0;                                                                 \
      t += gridDim.x;                                                         \
      if (t < total) fetch_bias(t);                                           \
    }                                                                         \
  }

  auto epilogue = [&](int tt) {
    int mtile, ntile, z;
    decode(tt, mtile, ntile, z);
    float* slab = g.out + (size_t)z * g.rows * g.ldo;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int row0 = mtile * BM + arow0 + mt * 32;
        const int col = ntile * BN + wcol0 + nt * 32 + r;
        const float bv = bias_reg[nt];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          float v = acc[mt][nt][q];
          if constexpr (F16) v = hilo(v, acc2[mt][nt][q]);
          acc[mt][nt][q] = 0.f;
          acc2[mt][nt][q] = 0.f;
          const int grow = row0 + acc_row(q, hh);
          if (grow >= g.rows) continue;
          if (EPI == 1) {
            slab[(size_t)grow * g.ldo + col] = v;
          } else {
            v += bv;
            if (g.act) v = gelu_tanh_fast(v);
            if (F16 && CLS == KC_GEMM_QKV) v = (fabsf(v) <= kF16Max) ? v : __builtin_nanf("");   // see gc_gemm_ws
            v = r16_if(v, g.round16);
            if (EPI == 0) g.out[(size_t)grow * g.ldo + col] = v;
            else store_s16(g.out, (size_t)grow, g.ldo, col, v);
          }
        }
      }
  };

  while (t < total) {
    // even step: LDS[0] holds position c, set 1 holds c+1 (if any), set 0 is free
    GC_STEP(ra0, rw0, ra1, rw1, have_next, 0);
    if (t >= total) break;
    // odd step: LDS[1] holds position c, set 0 holds c+1 (if refilled), set 1 is free
    GC_STEP(ra1, rw1, ra0, rw0, have_next, 1);
  }
#undef GC_LOAD
#undef GC_STAGE
#undef GC_STEP
}

template <int CLS>
static hipError_t launch_gemm_c(hipStream_t s, const GemmArgs& g_in, int shape, int splits, int epi, bool f16) {
  // shape 1: 32x128 tile, 2: 64x128 tile (256 threads)
  if (shape < 1 || shape > 2) return hipErrorInvalidValue;
  if (g_in.n % 128 || g_in.k_slice % kBK || g_in.lda % 4 || g_in.ldw % 4) return hipErrorInvalidValue;
  if (epi < 0 || epi > 1) return hipErrorInvalidValue;
  const int bm = shape == 1 ? 32 : 64;
  const int total = ((g_in.rows + bm - 1) / bm) * (g_in.n / 128) * splits;
  // Persistent grid: as many workgroups as can be co-resident (LDS-limited: 3 / 2 / 2 per CU),
  // and every workgroup gets the same number of output tiles.
  static int cap_override = -1;
  if (cap_override < 0) {
    const char* e = getenv("GC_TUNE_GEMM_BLOCKS");
    cap_override = (e && *e) ? atoi(e) : 0;
  }
  const int cap = cap_override > 0 ? cap_override : 256 * (shape == 1 ? 3 : 2);
  const int rounds = (total + cap - 1) / cap;
  int nblk = (total + rounds - 1) / rounds;
  if (rounds > 1) nblk = (nblk + 7) & ~7;     // keep blockIdx % 8 == tile % 8 for the XCD-aware order
  dim3 grid(nblk, 1, 1);
  GemmArgs g = g_in;
  g.splits = splits;
  if (g.att_S > 0) {                          // out-projection fed by attention partials (shape 1, slabs)
    if (shape != 1 || epi != 1 || g.att_S > kMaxAttnSplits) return hipErrorInvalidValue;
    if (f16) hipLaunchKernelGGL((gc_gemm_kernel<1, 4, 1, 1, 1, CLS, true, 1>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gc_gemm_kernel<1, 4, 1, 1, 1, CLS, false, 1>), grid, dim3(256), 0, s, g);
    return hipGetLastError();
  }
#define GC_LAUNCH(WM_, WN_, MT_, NT_, EPI_, F16_) \
  hipLaunchKernelGGL((gc_gemm_kernel<WM_, WN_, MT_, NT_, EPI_, CLS, F16_, 0>), grid, dim3(64 * WM_ * WN_), 0, s, g)
#define GC_SHAPES(EPI_, F16_)                                   \
  if (shape == 1) GC_LAUNCH(1, 4, 1, 1, EPI_, F16_);            \
  else GC_LAUNCH(1, 4, 2, 1, EPI_, F16_);
  if (epi == 0 && !f16) { GC_SHAPES(0, false) }
  else if (epi == 0 && f16) { GC_SHAPES(0, true) }
  else if (epi == 1 && !f16) { GC_SHAPES(1, false) }
  else if (epi == 1 && f16) { GC_SHAPES(1, true) }
  else return hipErrorInvalidValue;
#undef GC_SHAPES
#undef GC_LAUNCH
  return hipGetLastError();
}

// ----------------------------------------------------------------------------
// gc_gemm_ws: the f16x3 GEMM with the weight operand STREAMED straight into MFMA registers.
// With 1x4 waves per workgroup every wave owns its own 32 output columns, so a W fragment is
// used by exactly one wave: staging it through LDS (ds_write + ds_read + a barrier per K tile)
// buys nothing.  The weights are static, so gc_finalize lays them out in fragment order
// ("WF16": per (32-column tile, 16-deep k step) 1 KB of hi halfs then 1 KB of lo halfs, lane-
// major), which makes each fragment ONE fully coalesced 1-KB global_load_dwordx4 into the
// registers the MFMA reads.  A wave keeps kWsPD k-steps (16 KB) in flight with counted waits
// and never meets a barrier inside a K chunk.  The activation tile (BM rows x KC k values) is
// split to hi/lo once, staged in LDS per chunk and shared by the four waves; the next chunk's
// pieces are fetched into registers while the current chunk computes.
// The profile that motivated it (profiles/r01_*): the LDS-staged kernel's MFMA pipe was 12 %
// busy and a wave spent ~1.3 us per 32-deep K tile, i.e. it was L2-latency-bound with 20 KB in
// flight per workgroup.
// ----------------------------------------------------------------------------
#ifdef GC_STAMPS
// diagnostic builds (csrc/build.sh stamps, tools/stamp_gemm_ws.cpp): 10 words per wave of every gc_gemm_ws launch
__device__ unsigned long long* g_ws_stamps = nullptr;
hipError_t set_gemm_ws_stamp_buffer(unsigned long long* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_ws_stamps), &p, sizeof(p));
}
#define GC_WSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); wst[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define GC_WSTAMP_ACC(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t__ = __builtin_amdgcn_s_memtime(); wst[i] += t__ - wtp; wtp = t__; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define GC_WSTAMP(i) do { } while (0)
#define GC_WSTAMP_ACC(i) do { } while (0)
#endif

// A16: the A operand holds exact fp16 values ("fp16 node features": every producer rounds, so its lo plane is
// zero): the A-lo x W-hi MFMA and the lo-plane fragment read are left out -- 2 MFMAs per product, same bits.
// A separate instantiation (made by the second compilation of this file, -DGC_TU_A16), not a run-time branch:
// this kernel sits at its register budget and a branch around the product loop cost 100+ spilled registers
// (DESIGN.md section 3b).
// PIPE (software-pipelined activation staging; AMODE 0, 32- and 64-row tiles): the LDS tile is double-buffered in
// chunks of half the depth, and a wave stages chunk c + 1 in the same basic block as the MFMAs of chunk c (no
// branches) instead of in a phase of its own between two barriers: ONE barrier per chunk.
// Used by the gc_a16 build only (64-row tiles), where staging is a plain 16-byte copy: QKV -5 %, FFW-2 -6 %, node
// GEMMs -7 %, FFW-1 +-0 at the 1-degree size.  With float32 features the split arithmetic moves into the product
// loop and the loop grows by MORE than the phase it replaces (stamps, profiles/r03_stamps_gemm_ws_pipe.txt: loops
// 18.9k -> 28.1k cycles per wave for a 7.0k staging phase, wave lifetime 36.7k -> 39.8k): a SIMD's three waves do
// not hide one another's split VALU work under their MFMAs, so that form stays unpipelined (GC_TUNE_WS_PIPE=1
// forces it on for measurements).
// F32 (exact-f32 family): g.wt is the WF32 image, the LDS tile holds plain float32, a k16 step is 8
// v_mfma_f32_32x32x2_f32 into one accumulator (no PIPE / A16 / epilogue-3 forms).
template <int MT, int EPI, int CLS, int AMODE, int kWsPD /* W fragments (k16 steps) in flight per wave */,
          int OCC /* workgroups per CU the register budget is held to */, bool A16 = false, bool PIPE_ = false,
          bool F32 = false>
__global__ __launch_bounds__(256, OCC) void gc_gemm_ws_kernel(GemmArgs g) {
  static_assert(!F32 || (!A16 && !PIPE_ && EPI != 3), "exact-f32 form");
  constexpr bool PIPE = PIPE_ && AMODE == 0 && MT <= 2;
  // k values of one activation chunk in LDS (PIPE: the host guarantees k_slice % KC == 0, so kc == KC)
  constexpr int KC = PIPE ? (MT == 1 ? 128 : 64) : ((MT == 1) ? 256 : (MT == 2 ? 128 : 64));
  constexpr int NBUF = PIPE ? 2 : 1;
  constexpr int BM = 32 * MT, BN = 128;
  // transposed product (the weight fragment is the MFMA's A operand): the QKV epilogue, and every fp16-stored
  // output (A16, epi 0) -- a lane then owns 4 consecutive columns of a row and stores them as one 8-byte piece
  constexpr bool TR = EPI == 3 || (A16 && EPI == 0);
  constexpr int LDA = KC + 4;                 // 16-byte row shift: conflict-free ds_read_b128
  constexpr int AP = BM * (KC / 4) / 256;     // 16-byte activation pieces per thread per chunk (8)
  static_assert(AMODE == 0 || MT == 1, "attention-merging loader: 32-row tiles only");
  __shared__ __attribute__((aligned(16))) float Asb[NBUF][BM][LDA];
  float (*As)[LDA] = Asb[0];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int kc = PIPE ? KC : (g.k_slice < KC ? g.k_slice : KC);     // host: k_slice % kc == 0, kc % 64 == 0
  const int nchunks = g.k_slice / kc;
  const int ppr_lg = (kc == 256) ? 6 : (kc == 128 ? 5 : 4);   // log2(16-byte pieces per row per chunk)
  const int n_mtiles = (g.rows + BM - 1) / BM;
  const int n_ctiles = g.n / BN;
  const int n_panels = n_ctiles * g.splits;
  int mtile, ntile, z;
  {
    // Workgroups are dealt to the 8 XCDs round-robin (blockIdx.x % 8).  (row tile, panel) pairs are
    // numbered row-tile-major and every XCD takes a contiguous, balanced range: the panels of one
    // row tile then run on ONE XCD, whose L2 serves the activation tile to all but the first of
    // them, while each XCD streams the whole (small) weight matrix once.  With panels spread over
    // the XCDs instead, every XCD read every activation tile (QKV: 22 vs 9 MB from the Infinity Cache).
    const int n_pairs = n_mtiles * n_panels;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int base_cnt = n_pairs >> 3, extra = n_pairs & 7;
    if (j >= base_cnt + (xcd < extra ? 1 : 0)) return;
    const int lin = xcd * base_cnt + (xcd < extra ? xcd : extra) + j;
    mtile = lin / n_panels;
    const int panel = lin - mtile * n_panels;
    ntile = panel % n_ctiles;
    z = panel / n_ctiles;
  }
  const int kbase = z * g.k_slice;
#ifdef GC_STAMPS
  unsigned long long wst[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, wtp = 0;
  wst[8] = __builtin_amdgcn_s_memrealtime();   // constant 100 MHz: gives the shader clock the launch ran at
#endif
  GC_WSTAMP(0);

  // ---- W stream of this wave: fragments of column tile (ntile*4 + wave), k16 steps of slice z
  const int nsteps = g.k_slice / 16;
  const float* wf = g.wt + ((size_t)(ntile * 4 + wave) * (g.ldw / 16) + (size_t)z * nsteps) * 512 + lane * 4;
  f32x4 wh[kWsPD], wl[kWsPD];
#pragma unroll
  for (int i = 0; i < kWsPD; ++i) {
    const int sn = i < nsteps ? i : nsteps - 1;
    wh[i] = ld4(wf + (size_t)sn * 512);
    wl[i] = ld4(wf + (size_t)sn * 512 + 256);
  }
  const float bias_reg = (EPI != 1 && g.bias) ? g.bias[ntile * BN + wave * 32 + r] : 0.f;

  // ---- activation pieces: piece p = tid + 256 i  ->  (row p >> ppr_lg, 16-byte column p & mask)
  f32x4 ra[AP];
  auto piece_src = [&](int i, int& row, int& c4) {
    const int p = tid + 256 * i;
    row = p >> ppr_lg;
    c4 = p & ((1 << ppr_lg) - 1);
    const int grow = mtile * BM + row;
    return grow < g.rows ? grow : g.rows - 1;
  };
  auto load_chunk = [&](int c) {              // AMODE 0: next chunk's pieces into registers
    if constexpr (A16) {                       // A is stored as halfs: 16-byte pieces of 8 k values, half as many
#pragma unroll
      for (int i = 0; i < AP / 2; ++i) {
        const int p = tid + 256 * i;
        const int row = p >> (ppr_lg - 1), c8 = p & ((1 << (ppr_lg - 1)) - 1);
        int grow = mtile * BM + row;
        if (grow >= g.rows) grow = g.rows - 1;
        if (row < BM) ra[i] = ld4(reinterpret_cast<const float*>(as_h16(g.a) + (size_t)grow * g.lda + kbase + c * kc + c8 * 8));
      }
      asm volatile("" ::: "memory");
      return;
    }
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      int row, c4;
      const int grow = piece_src(i, row, c4);
      if (row < BM) ra[i] = ld4(g.a + (size_t)grow * g.lda + kbase + c * kc + c4 * 4);
    }
    // a compiler-level memory fence, so that LLVM cannot sink these loads towards their first use in stage_chunk
    // one chunk later (measured neutral: the staging phase is bound by the split arithmetic, not by the loads)
    asm volatile("" ::: "memory");
  };
  // AMODE 1: merge the attention key-split partials (see gc_gemm_kernel) and stage them, four
  // pieces at a time (each piece holds up to 4 x 6 registers of partials while in flight)
  auto fill_chunk_att = [&](int c) {
#pragma unroll 4
    for (int i = 0; i < AP; ++i) {
      int row, c4;
      const int grow = piece_src(i, row, c4);
      if (row >= BM) break;
      const int col = kbase + c * kc + c4 * 4;
      const int node = grow / g.att_B, bb = grow - node * g.att_B;
      const int head = col / g.att_DH, dv = col - head * g.att_DH;
      const int q = node % kTileM;
      const size_t slot0 = ((size_t)(node / kTileM) * g.att_S * g.att_B + bb) * g.att_H + head;
      f32x4 po[kMaxAttnSplits];
      float pm[kMaxAttnSplits], pl[kMaxAttnSplits];
#pragma unroll
      for (int sp = 0; sp < kMaxAttnSplits; ++sp)
        if (sp < g.att_S) {
          const size_t slot = slot0 + (size_t)sp * g.att_B * g.att_H;
          po[sp] = ld4(g.att_po + slot * (kTileM * g.att_DH) + q * g.att_DH + dv);
          pm[sp] = g.att_pml[slot * (kTileM * 2) + q * 2];
          pl[sp] = g.att_pml[slot * (kTileM * 2) + q * 2 + 1];
        }
      float mstar = -1e30f;
#pragma unroll
      for (int sp = 0; sp < kMaxAttnSplits; ++sp)
        if (sp < g.att_S) mstar = fmaxf(mstar, pm[sp]);
      f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
      float lsum = 0.f;
#pragma unroll
      for (int sp = 0; sp < kMaxAttnSplits; ++sp)
        if (sp < g.att_S) {
          const float w = (pl[sp] != 0.f) ? __expf(pm[sp] - mstar) : 0.f;
          acc4 += po[sp] * w;
          lsum += w * pl[sp];
        }
      if constexpr (F32) st4(&As[row][c4 * 4], acc4 * ((lsum != 0.f) ? 1.0f / lsum : 0.f));
      else stage16<A16>(&As[row][(c4 >> 3) * 32], c4 & 7, r16_if(acc4 * ((lsum != 0.f) ? 1.0f / lsum : 0.f), g.round16));
    }
  };
  auto stage_chunk = [&]() {
    if constexpr (A16) {                       // halfs as loaded: group c8 / 4 of the row, 16 bytes into its hi plane
#pragma unroll
      for (int i = 0; i < AP / 2; ++i) {
        const int p = tid + 256 * i;
        const int row = p >> (ppr_lg - 1), c8 = p & ((1 << (ppr_lg - 1)) - 1);
        if (row < BM) st4(&As[row][(c8 >> 2) * 32 + (c8 & 3) * 4], ra[i]);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const int p = tid + 256 * i;
      const int row = p >> ppr_lg, c4 = p & ((1 << ppr_lg) - 1);
      if constexpr (F32) { if (row < BM) st4(&As[row][c4 * 4], ra[i]); }
      else if (row < BM) stage16<A16>(&As[row][(c4 >> 3) * 32], c4 & 7, ra[i]);
    }
  };

  f32x16 acc[MT], acc2[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      acc[mt][q] = 0.f;
      acc2[mt][q] = 0.f;
    }

  int s = 0;                                   // W step about to be consumed
  if constexpr (PIPE) {
    // Two register sets of AP pieces: chunk c + 1 sits in one (loaded a whole chunk earlier) while chunk c + 2
    // lands in the other.  Every piece is in range (BM * KC / 4 == 256 AP exactly) and chunk indices are clamped,
    // so the chunk body below is branch-free.
    constexpr int PPR_LG = KC == 128 ? 5 : 4;
    constexpr int NP = A16 ? AP / 2 : AP;      // 16-byte pieces per thread per chunk (A16: 8 halfs each)
    constexpr int PL = A16 ? PPR_LG - 1 : PPR_LG;
    f32x4 rb[AP];
    auto pload = [&](int c, f32x4 (&rs)[AP]) {
      const int cc = c < nchunks ? c : nchunks - 1;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int p = tid + 256 * i;
        const int row = p >> PL, cp = p & ((1 << PL) - 1);
        int grow = mtile * BM + row;
        if (grow >= g.rows) grow = g.rows - 1;
        if constexpr (A16) rs[i] = ld4(reinterpret_cast<const float*>(as_h16(g.a) + (size_t)grow * g.lda + kbase + cc * KC + cp * 8));
        else rs[i] = ld4(g.a + (size_t)grow * g.lda + kbase + cc * KC + cp * 4);
      }
      asm volatile("" ::: "memory");
    };
    auto pstage = [&](float (*dst)[LDA], f32x4 (&rs)[AP]) {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int p = tid + 256 * i;
        const int row = p >> PL, cp = p & ((1 << PL) - 1);
        if constexpr (A16) st4(&dst[row][(cp >> 2) * 32 + (cp & 3) * 4], rs[i]);
        else stage16<false>(&dst[row][(cp >> 3) * 32], cp & 7, rs[i]);
      }
    };
    pload(0, ra);
    pload(1, rb);
    GC_WSTAMP(1);
#ifdef GC_STAMPS
    wtp = wst[1];
#endif
    pstage(Asb[0], ra);
    GC_WSTAMP_ACC(3);
    __syncthreads();
    GC_WSTAMP_ACC(4);
    pload(2, ra);
    // chunk c: buffer `cur` is read; `rs` (chunk c + 1) is split and staged into `nxt` among the first MFMAs, then
    // refilled with chunk c + 3; ONE barrier ends the chunk (cur released, nxt complete).  After the last chunk the
    // staged bytes are a clamped re-read nobody uses.
    auto chunk = [&](int c, float (*cur)[LDA], float (*nxt)[LDA], f32x4 (&rs)[AP]) __attribute__((always_inline)) {
#pragma unroll
      for (int ks = 0; ks < KC / 16; ks += kWsPD) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          if (ks == 0 && half == 0) pstage(nxt, rs);
#pragma unroll
          for (int j = 0; j < kWsPD / 2; ++j) {
            const int i = half * (kWsPD / 2) + j;
            const int kk = ks + i;
            const int off = (kk >> 1) * 32 + (kk & 1) * 8 + hh * 4;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              const f32x4 ah = ld4(&cur[mt * 32 + r][off]);
              f32x4 al;
              if constexpr (!A16) al = ld4(&cur[mt * 32 + r][off + 16]);
              if constexpr (TR) {
                acc2[mt] = mfma16(wl[i], ah, acc2[mt]);
                acc[mt] = mfma16(wh[i], ah, acc[mt]);
                if constexpr (!A16) acc2[mt] = mfma16(wh[i], al, acc2[mt]);
              } else {
                acc2[mt] = mfma16(ah, wl[i], acc2[mt]);
                acc[mt] = mfma16(ah, wh[i], acc[mt]);
                if constexpr (!A16) acc2[mt] = mfma16(al, wh[i], acc2[mt]);
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < kWsPD / 2; ++j) {
            const int i = half * (kWsPD / 2) + j;
            int sn = s + kWsPD + j;
            if (sn >= nsteps) sn = nsteps - 1;
            wh[i] = ld4(wf + (size_t)sn * 512);
            wl[i] = ld4(wf + (size_t)sn * 512 + 256);
          }
          if (ks == 0 && half == 0) pload(c + 3, rs);
          s += kWsPD / 2;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      GC_WSTAMP_ACC(5);
      __syncthreads();
      GC_WSTAMP_ACC(2);
    };
    for (int c = 0; c < nchunks; c += 2) {
      chunk(c, Asb[0], Asb[NBUF - 1], rb);
      if (c + 1 < nchunks) chunk(c + 1, Asb[NBUF - 1], Asb[0], ra);
    }
  } else {
  if constexpr (AMODE == 0) load_chunk(0);
  GC_WSTAMP(1);                                // W ring and the first A chunk issued
#ifdef GC_STAMPS
  wtp = wst[1];
#endif
  for (int c = 0; c < nchunks; ++c) {
    if (c) __syncthreads();                    // every wave is done reading the previous chunk
    GC_WSTAMP_ACC(2);                          // (sum) barrier: previous chunk released
    if constexpr (AMODE == 0) stage_chunk();
    else fill_chunk_att(c);
    GC_WSTAMP_ACC(3);                          // (sum) wait for the A pieces + split + LDS writes
    __syncthreads();
    GC_WSTAMP_ACC(4);                          // (sum) barrier: chunk staged
    if constexpr (AMODE == 0)
      if (c + 1 < nchunks) load_chunk(c + 1);  // lands while this chunk computes
    // The ring is consumed and refilled in two halves, so that half a ring of loads is always
    // in flight behind the MFMAs of the other half (the scheduling barriers keep hipcc from
    // sinking every refill to the end of the group, which would expose a full L2 round trip).
    for (int ks = 0; ks < kc / 16; ks += kWsPD) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int j = 0; j < kWsPD / 2; ++j) {
          const int i = half * (kWsPD / 2) + j;
          const int kk = ks + i;               // S16 row: group kk/2 = [32 hi | 32 lo], k16 sub-step kk&1
          const int off = F32 ? kk * 16 + hh * 8 : (kk >> 1) * 32 + (kk & 1) * 8 + hh * 4;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            if constexpr (F32) {                 // float32 row: the lane's 8 consecutive k of the step; weight halves in wh / wl
              const f32x4 a0 = ld4(&As[mt * 32 + r][off]), a1 = ld4(&As[mt * 32 + r][off + 4]);
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[mt] = TR ? mfma32(wh[i][e], a0[e], acc[mt]) : mfma32(a0[e], wh[i][e], acc[mt]);
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[mt] = TR ? mfma32(wl[i][e], a1[e], acc[mt]) : mfma32(a1[e], wl[i][e], acc[mt]);
              continue;
            }
            const f32x4 ah = ld4(&As[mt * 32 + r][off]);
            f32x4 al;
            if constexpr (!A16) al = ld4(&As[mt * 32 + r][off + 16]);
            if constexpr (TR) {                      // transposed product: a lane gets 4 consecutive columns of row r
              acc2[mt] = mfma16(wl[i], ah, acc2[mt]);
              acc[mt] = mfma16(wh[i], ah, acc[mt]);
              if constexpr (!A16) acc2[mt] = mfma16(wh[i], al, acc2[mt]);
            } else {
              acc2[mt] = mfma16(ah, wl[i], acc2[mt]);  // the two MFMAs into acc2 are kept apart
              acc[mt] = mfma16(ah, wh[i], acc[mt]);
              if constexpr (!A16) acc2[mt] = mfma16(al, wh[i], acc2[mt]);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < kWsPD / 2; ++j) {
          const int i = half * (kWsPD / 2) + j;
          int sn = s + kWsPD + j;              // refill this ring slot (clamped: harmless re-read)
          if (sn >= nsteps) sn = nsteps - 1;
          wh[i] = ld4(wf + (size_t)sn * 512);
          wl[i] = ld4(wf + (size_t)sn * 512 + 256);
        }
        s += kWsPD / 2;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    GC_WSTAMP_ACC(5);                          // (sum) product loop of the chunk
  }
  }
#ifdef GC_STAMPS
  auto stamp_out = [&]() {
    GC_WSTAMP(6);                              // epilogue stores issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GC_WSTAMP(7);
    wst[9] = __builtin_amdgcn_s_memrealtime();
    if (g_ws_stamps && lane == 0) {
      unsigned long long* o = g_ws_stamps + ((size_t)blockIdx.x * 4 + wave) * 10;
      for (int i = 0; i < 10; ++i) o[i] = wst[i];
    }
  };
#endif

  // Epilogue.  The variants are separated up front (whole tile in range or not, activation or
  // not) so that the common case is 16 unconditional stores per accumulator tile: with per-row
  // branches hipcc puts a conservative vmcnt wait in front of every store and serialises them.
  if constexpr (EPI == 3) {
    // QKV projection (no bias): register q of lane (r, hh) = column acc_row(q, hh) of this wave's 32-column
    // tile, row r.  q and the f32 copies of k, v go out as 16-byte stores; k and v additionally as fp16
    // hi / lo planes (8-byte stores) so that the attention kernel never splits them again.
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    const int D = g.kv_d;
    const int gcol = ntile * BN + wave * 32;          // first column of this wave's tile in [0, 3D)
    const int which = gcol / D, col_in = gcol - which * D;
    _Float16* kv = reinterpret_cast<_Float16*>(g.kv16);
    with_flag(g.round16, [&](auto rc) __attribute__((always_inline)) {
    constexpr bool RND = decltype(rc)::value;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int grow = mtile * BM + mt * 32 + r;
      if (grow >= g.rows) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = hilo(acc[mt][4 * j + e], acc2[mt][4 * j + e]);
          x = (fabsf(x) <= kF16Max) ? x : __builtin_nanf("");     // leaves the f16x3 domain here or never
          v[e] = r16_c<RND>(x);
        }
        const int c = col_in + 8 * j + 4 * hh;
        if (which == 0) {                       // q: float32 (the attention kernel splits it once per tile)
          if constexpr (A16) sth4(as_h16(g.out) + (size_t)grow * g.ldo + c, v);   // fp16 storage: halfs, row stride ldo
          else st4(g.out + (size_t)grow * g.ldo + c, v);
        } else if constexpr (A16) {             // fp16 storage: k, v are their own hi planes (the lo planes stay unwritten, unread)
          sth4(kv + (size_t)grow * (4 * D) + (size_t)(which - 1) * (2 * D) + c, v);
        } else {                                // k, v: only the planes (gc_debug_fetch rebuilds float32 from them)
          _Float16 h4[4], l4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) split16(v[e], h4[e], l4[e]);
          _Float16* dst = kv + (size_t)grow * (4 * D) + (size_t)(which - 1) * (2 * D) + c;
          *reinterpret_cast<f16x4*>(dst) = f16x4{h4[0], h4[1], h4[2], h4[3]};
          *reinterpret_cast<f16x4*>(dst + D) = f16x4{l4[0], l4[1], l4[2], l4[3]};
        }
      }
    }
    });
#ifdef GC_STAMPS
    stamp_out();
#endif
    return;
  }
  if constexpr (A16 && EPI == 0) {
    // fp16-stored output (FFW layer 1 at d_model 512): transposed product, so lane (r, hh) holds columns
    // 8 j + 4 hh + (0..3) of row r of this wave's 32-column tile: bias (+ gelu), then one 8-byte store per group
    const int cb = ntile * BN + wave * 32 + 4 * hh;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 bv = g.bias ? ld4(g.bias + cb + 8 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int grow = mtile * BM + mt * 32 + r;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = hilo(acc[mt][4 * j + e], acc2[mt][4 * j + e]) + bv[e];
          if (g.act) x = gelu_tanh_fast(x);
          v[e] = x;
        }
        if (grow < g.rows) {
          if (g.out_f32) st4(g.out + (size_t)grow * g.ldo + cb + 8 * j, v);     // pre-activation terms: float32, unrounded
          else sth4(as_h16(g.out) + (size_t)grow * g.ldo + cb + 8 * j, v);
        }
      }
    }
#ifdef GC_STAMPS
    stamp_out();
#endif
    return;
  }
  float bias_v = bias_reg;
  asm volatile("" : "+v"(bias_v));            // the bias load is waited for here, once
  float* obase = (EPI == 1 ? g.out + (size_t)z * g.rows * g.ldo : g.out) + ntile * BN + wave * 32 + r;
  auto emit = [&](auto full_c, auto act_c, auto rnd_c) {
    constexpr bool FULL = decltype(full_c)::value, ACT = decltype(act_c)::value, RND = decltype(rnd_c)::value;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row0 = mtile * BM + mt * 32;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        float v = hilo(acc[mt][q], acc2[mt][q]);
        const int grow = row0 + acc_row(q, hh);
        if (EPI != 1) {
          v += bias_v;
          if (ACT) v = gelu_tanh_fast(v);
        }
        // Q, K, V leave the f16x3 domain here or never: attention splits them without a check
        if (!F32 && CLS == KC_GEMM_QKV && EPI == 0) v = (fabsf(v) <= kF16Max) ? v : __builtin_nanf("");
        if (EPI != 1) v = r16_c<RND>(v);
        if (FULL || grow < g.rows) obase[(size_t)grow * g.ldo] = v;
      }
    }
  };
  const bool full = mtile * BM + BM <= g.rows;
  const bool act = EPI != 1 && g.act;
  if (EPI != 1 && g.round16) {                 // fp16-feature mode (rare): one generic variant
    if (act) emit(std::false_type{}, std::true_type{}, std::true_type{});
    else emit(std::false_type{}, std::false_type{}, std::true_type{});
  } else if (full && act) emit(std::true_type{}, std::true_type{}, std::false_type{});
  else if (full) emit(std::true_type{}, std::false_type{}, std::false_type{});
  else if (act) emit(std::false_type{}, std::true_type{}, std::false_type{});
  else emit(std::false_type{}, std::false_type{}, std::false_type{});
#ifdef GC_STAMPS
  stamp_out();
#endif
}

template <int CLS>
static hipError_t launch_gemm_ws_c(hipStream_t s, const GemmArgs& g_in, int mt, int splits, int epi) {
  if ((mt != 1 && mt != 2 && mt != 4) || epi < 0 || (epi > 1 && epi != 3)) return hipErrorInvalidValue;
  if (epi == 3 && (CLS != KC_GEMM_QKV || !g_in.kv16 || g_in.kv_d % 32 || g_in.n != 3 * g_in.kv_d || splits != 1 ||
                   g_in.att_S > 0 || g_in.bias))
    return hipErrorInvalidValue;
  const int KC = (mt == 1) ? 256 : (mt == 2 ? 128 : 64);
  const int kc = g_in.k_slice < KC ? g_in.k_slice : KC;
  if (g_in.n % 128 || kc % 64 || g_in.k_slice % 128 || g_in.k_slice % kc || g_in.lda % 4 || g_in.ldw % 16 || splits < 1)
    return hipErrorInvalidValue;
  GemmArgs g = g_in;
  g.splits = splits;
  const int BM = 32 * mt;
  const int total = ((g.rows + BM - 1) / BM) * (g.n / 128) * splits;
  if (total <= 0) return hipSuccess;
  dim3 grid((total + 7) / 8 * 8), block(256);   // padded: the kernel maps XCD-contiguous ranges
  // ring of 4 k16 steps, registers held to 3 workgroups per CU: the best of {ring 8 / 2 per CU,
  // ring 4 / 4 (spills), ring 4 / 3, ring 8 / 3 (spills)} at the nano shapes (tools/bench_kernels ws)
  static int pipe_env = -1;
  if (pipe_env < 0) {
    const char* e = getenv("GC_TUNE_WS_PIPE");
    pipe_env = (e && *e) ? (atoi(e) != 0) : 2;   // 2: the default rule
  }
  const bool pipe = pipe_env == 2 ? (kTuA16 && mt == 2) : pipe_env != 0;
#define GC_WS(MT_, EPI_, AM_)                                                                                         \
  {                                                                                                                   \
    if (pipe && AM_ == 0 && MT_ <= 2)                                                                                 \
      hipLaunchKernelGGL((gc_gemm_ws_kernel<MT_, EPI_, CLS, AM_, 4, (MT_ >= 4 ? 2 : 3), kTuA16, (AM_ == 0 && MT_ <= 2)>), \
                         grid, block, 0, s, g);                                                                       \
    else                                                                                                              \
      hipLaunchKernelGGL((gc_gemm_ws_kernel<MT_, EPI_, CLS, AM_, 4, (MT_ >= 4 ? 2 : 3), kTuA16, false>), grid, block, 0, s, g); \
  }
  if constexpr (!kTuA16) {
    if (g.f32w) {                                // exact-f32 family: plain tiles only (32 / 64 rows), epilogues 0 and 1
#define GC_WS32(MT_, EPI_, AM_) hipLaunchKernelGGL((gc_gemm_ws_kernel<MT_, EPI_, CLS, AM_, 4, 3, false, false, true>), grid, block, 0, s, g);
      if (epi == 3 || mt == 4) return hipErrorInvalidValue;
      if (g.att_S > 0) {
        if (mt != 1 || epi != 1 || g.att_S > kMaxAttnSplits) return hipErrorInvalidValue;
        GC_WS32(1, 1, 1)
      } else if (mt == 1 && epi == 0) { GC_WS32(1, 0, 0) }
      else if (mt == 1) { GC_WS32(1, 1, 0) }
      else if (epi == 0) { GC_WS32(2, 0, 0) }
      else { GC_WS32(2, 1, 0) }
#undef GC_WS32
      return hipGetLastError();
    }
  }
  if (g.att_S > 0) {
    if (mt != 1 || epi != 1 || g.att_S > kMaxAttnSplits) return hipErrorInvalidValue;
    GC_WS(1, 1, 1)
  } else if (epi == 3) {
    if constexpr (CLS == KC_GEMM_QKV) {
      if (mt == 1) { GC_WS(1, 3, 0) } else if (mt == 2) { GC_WS(2, 3, 0) } else { GC_WS(4, 3, 0) }
    }
  } else if (mt == 1 && epi == 0) { GC_WS(1, 0, 0) }
  else if (mt == 1 && epi == 1) { GC_WS(1, 1, 0) }
  else if (mt == 2 && epi == 0) { GC_WS(2, 0, 0) }
  else if (mt == 2) { GC_WS(2, 1, 0) }
  else if (epi == 0) { GC_WS(4, 0, 0) }
  else { GC_WS(4, 1, 0) }
#undef GC_WS
  return hipGetLastError();
}

// ----------------------------------------------------------------------------
// gc_gemm_rowop: a full-width projection with the residual stream's row pass in its epilogue
// (the attention out-projection followed by "x += bias + y; h = cond(LayerNorm(x))").
// A workgroup owns 32 whole rows: 8 waves x NT 32-column tiles cover all n = d_model columns and
// the whole K, so the finished rows are in hand and no slab round trip or second launch is
// needed.  Only rows/32 workgroups exist (81 at the nano size), but the split-K form was already
// at the launch-latency floor and its row pass cost another launch of the same length.
// ----------------------------------------------------------------------------
// RH = rows a workgroup finishes: 32, or 16 when there are too few 32-row tiles to occupy half the
// chip -- the MFMA tile stays 32 rows high (its upper half computes garbage nobody reads: MFMA time is
// negligible here) but twice as many CUs share the merge loads and the row pass, which are the bulk.
#ifdef GC_STAMPS
__device__ unsigned long long* g_rowop_stamps = nullptr;   // diagnostic builds (tools/stamp_rowop.cpp): 8 words per wave
hipError_t set_gemm_rowop_stamp_buffer(unsigned long long* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_rowop_stamps), &p, sizeof(p));
}
#define GC_RSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); rst[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define GC_RSTAMP(i) do { } while (0)
#endif

// RH = 48 (rounds of workgroups): a workgroup is alone on its CU (8 waves x ~200 registers), so rows / 32 tiles run as
// ceil(tiles / 256) ROUNDS and a partly filled last round costs as much as a full one.  At the 1-degree size 321
// 32-row tiles are a round of 256 plus one of 65; 214 workgroups of 48 rows are ONE round, each streaming W once for
// 1.5x the rows (two MFMA row tiles, the upper half of the second computing garbage nobody reads).
template <int NT, int AMODE, int CLS, int RH, bool A16 = false /* exact-fp16 A: 2 MFMAs per product */,
          bool F32 = false /* exact-f32 family: WF32 image, float32 A tile, v_mfma_f32_32x32x2_f32 */>
__global__ __launch_bounds__(512, 1) void gc_gemm_rowop_kernel(GemmArgs g, RowFuse f) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // A tile [32 MT][D+4] (S16), later y [32 MT][D+4]
  constexpr int MT = RH > 32 ? 2 : 1;          // MFMA row tiles per workgroup
  constexpr int R = NT == 1 ? 8 : 4;
  const int D = g.n, LDA = D + 4;
  const int tid = threadIdx.x, nthr = blockDim.x, nwave = nthr >> 6;
  const int wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int mtile = blockIdx.x;
#ifdef GC_STAMPS
  unsigned long long rst[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long rrt0 = __builtin_amdgcn_s_memrealtime();
#endif
  GC_RSTAMP(0);
  const int steps = D / 16;
  const float* wf = g.wt + (size_t)(wave * NT) * steps * 512 + lane * 4;
  const size_t cts = (size_t)steps * 512;
  f32x4 wh[R][NT], wl[R][NT];
  ws_ring_fill<NT, R>(wh, wl, wf, cts, steps);

  // Operands of the row pass at the END of the kernel, fetched now: this wave's first four residual rows, the
  // bias and (one batch element) the conditioning scale / offset of this lane's columns.  Issued down there
  // they were a dependent round trip to data the previous layer left on other XCDs -- the row pass took
  // 5.7 k of a wave's 18.9 k cycles (tools/stamp_rowop.cpp), 4.2 k of 16.4 k with the loads up here
  // (11.1 -> 10.1 us per launch back to back).
  auto load_x_rows = [&](int rb, f32x4 (&v)[4][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int rr = rb + k * nwave;
      int row = mtile * RH + rr;
      if (row >= g.rows) row = g.rows - 1;      // clamped rows are computed but not stored
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = 4 * lane + 256 * i;
        if constexpr (A16)                      // fp16 activation storage: the residual stream is halfs
          v[k][i] = (c < D && rr < RH) ? ldh4(as_h16(f.x) + (size_t)row * D + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        else
          v[k][i] = (c < D && rr < RH) ? ld4(f.x + (size_t)row * D + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  // (only in the 16-row form, i.e. when few row tiles exist and every kernel is latency-bound: at the 1-degree
  //  size -- 32-row tiles, d_model 512 -- the early loads made the kernel 7 % slower, 49.0 -> 52.7 us)
  constexpr bool kEarly = RH == 16;
  f32x4 xpre[4][2], bpre[2], scpre[2], ofpre[2];
  if constexpr (kEarly) {
    load_x_rows(wave, xpre);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = 4 * lane + 256 * i;
      const bool in = c < D;
      bpre[i] = (in && f.bias) ? ld4(f.bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      scpre[i] = (in && f.B == 1) ? ld4(f.cond + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      ofpre[i] = (in && f.B == 1) ? ld4(f.cond + D + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }

  // ---- A tile: 32 rows x D, split to hi/lo on the way into LDS ----
  const int ppr = D / 4;                       // 16-byte pieces per row
  // up to four pieces per thread per pass, unrolled so that their loads (up to 3 x 4 per piece when
  // merging partials) are all in flight together; RH * ppr is a multiple of UP * nthr
  constexpr int UP = (RH * NT / 8) < 4 ? (RH * NT / 8) : 4;
  if constexpr (AMODE == 0 && A16) {          // A (the attention output) is stored as halfs: 16-byte pieces, copied as they are
    const int ppr8 = D / 8;
    for (int p = tid; p < RH * ppr8; p += nthr) {
      const int row = p / ppr8, c8 = p - row * ppr8;
      int grow = mtile * RH + row;
      if (grow >= g.rows) grow = g.rows - 1;
      st4(smem + row * LDA + (c8 >> 2) * 32 + (c8 & 3) * 4,
          ld4(reinterpret_cast<const float*>(as_h16(g.a) + (size_t)grow * g.lda + c8 * 8)));
    }
  } else if constexpr (AMODE == 0) {
    for (int p0 = tid; p0 < RH * ppr; p0 += UP * nthr)
#pragma unroll
      for (int pi = 0; pi < UP; ++pi) {
        const int p = p0 + pi * nthr;
        const int row = p / ppr, c4 = p - row * ppr;
        int grow = mtile * RH + row;
        if (grow >= g.rows) grow = g.rows - 1;
        if constexpr (F32) st4(smem + row * LDA + c4 * 4, ld4(g.a + (size_t)grow * g.lda + c4 * 4));
        else stage16<A16>(smem + row * LDA + (c4 >> 3) * 32, c4 & 7, ld4(g.a + (size_t)grow * g.lda + c4 * 4));
      }
  }
  if constexpr (AMODE == 0) {
    // Rows of attention tiles that the item-list launch cut into key-range pieces (g.att_tiles): merged from the pieces'
    // partial (m, l, O) triples and staged over what the loop above put there.  A wave's 64 pieces lie in one row, so
    // the test is wave-uniform; at most kItemPieces pieces per tile (loads of a missing piece re-read the last one).
    if (g.att_tiles) {                          // (item lists exist for batch 1 only: row = mesh node)
      constexpr int N = kItemPieces, PB = 4;    // PB pieces x (N partial rows + N (m, l) pairs) in flight per thread
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      const int row_lo = mtile * RH, row_hi = (row_lo + RH < g.rows) ? row_lo + RH : g.rows;
      for (int tt = row_lo / kTileM; tt * kTileM < row_hi; ++tt) {
        const int np = g.att_tiles[2 * tt + 1];
        if (np == 0) continue;                  // workgroup-uniform
        const int slot_first = g.att_tiles[2 * tt];
        __syncthreads();                        // the loop above staged these rows too, from other threads (workgroup-uniform branch)
        const int r0 = (tt * kTileM > row_lo ? tt * kTileM : row_lo), r1 = ((tt + 1) * kTileM < row_hi ? (tt + 1) * kTileM : row_hi);
        const int npc = (r1 - r0) * ppr;        // a multiple of 32 * ppr / ... pieces of this tile's rows inside the workgroup
        for (int p0 = tid; p0 < npc; p0 += PB * nthr) {
          f32x4 po[PB][N];
          f32x2 ml[PB][N];
#pragma unroll
          for (int pi = 0; pi < PB; ++pi) {
            int p = p0 + pi * nthr;
            if (p >= npc) p = npc - 1;          // clamped duplicate (staged twice with the same value)
            const int rr = p / ppr, c4 = p - rr * ppr;
            const int node = r0 + rr, col = c4 * 4;
            const int head = col / g.att_DH, dv = col - head * g.att_DH;
            const int q = node - tt * kTileM;
#pragma unroll
            for (int sp = 0; sp < N; ++sp) {
              const int spc = sp < np ? sp : np - 1;
              const size_t slot = (size_t)(slot_first + spc) * g.att_H + head;
              po[pi][sp] = ld4(g.att_po + slot * (kTileM * g.att_DH) + q * g.att_DH + dv);
              ml[pi][sp] = *reinterpret_cast<const f32x2*>(g.att_pml + slot * (kTileM * 2) + q * 2);
            }
          }
#pragma unroll
          for (int pi = 0; pi < PB; ++pi) {
            int p = p0 + pi * nthr;
            if (p >= npc) p = npc - 1;
            const int rr = p / ppr, c4 = p - rr * ppr;
            const int row = r0 + rr - row_lo;
            float mstar = -1e30f;
#pragma unroll
            for (int sp = 0; sp < N; ++sp) mstar = fmaxf(mstar, sp < np ? ml[pi][sp][0] : -1e30f);
            f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
            float lsum = 0.f;
#pragma unroll
            for (int sp = 0; sp < N; ++sp) {
              const float w = (sp < np && ml[pi][sp][1] != 0.f) ? __expf(ml[pi][sp][0] - mstar) : 0.f;
              acc4 += po[pi][sp] * w;
              lsum += w * ml[pi][sp][1];
            }
            f32x4 v = acc4 * ((lsum != 0.f) ? 1.0f / lsum : 0.f);
            if (f.round16) v = r16_c<true>(v);
            if constexpr (F32) st4(smem + row * LDA + c4 * 4, v);          // exact-f32 family: plain float32 tile
            else stage16<A16>(smem + row * LDA + (c4 >> 3) * 32, c4 & 7, v);
          }
        }
      }
    }
  }
  if constexpr (AMODE != 0) {
    // Merge of the attention key-split partials (see gc_gemm_kernel).  N = compile-time bound on the splits: the
    // loads of all pieces and splits of a pass are issued unconditionally first (a split beyond att_S re-reads
    // the last real one, an L1 hit, and gets weight 0), then merged.  (With the loads inside `if (sp < att_S)`
    // blocks hipcc put a full s_waitcnt between one piece's loads and the next piece's.  The phase is 6.1 k of a
    // wave's 16.4 k cycles either way -- one round trip to data the attention kernel has just written on other
    // XCDs plus ~2 k cycles of merge arithmetic and fp16 splits; tools/stamp_rowop.cpp.)
    auto merge_pass = [&](auto nmax) __attribute__((always_inline)) {
      constexpr int N = decltype(nmax)::value;
      constexpr int PB = (N * UP <= 8) ? UP : (N <= 4 ? 2 : 1);   // pieces whose loads fly together (registers)
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      for (int p0 = tid; p0 < RH * ppr; p0 += PB * nthr) {
        f32x4 po[PB][N];
        f32x2 ml[PB][N];
#pragma unroll
        for (int pi = 0; pi < PB; ++pi) {
          const int p = p0 + pi * nthr;
          const int row = p / ppr, c4 = p - row * ppr;
          int grow = mtile * RH + row;
          if (grow >= g.rows) grow = g.rows - 1;
          const int col = c4 * 4;
          const int node = grow / g.att_B, bb = grow - node * g.att_B;
          const int head = col / g.att_DH, dv = col - head * g.att_DH;
          const int q = node % kTileM;
          const size_t slot0 = ((size_t)(node / kTileM) * g.att_S * g.att_B + bb) * g.att_H + head;
#pragma unroll
          for (int sp = 0; sp < N; ++sp) {
            const int spc = sp < g.att_S ? sp : g.att_S - 1;
            const size_t slot = slot0 + (size_t)spc * g.att_B * g.att_H;
            po[pi][sp] = ld4(g.att_po + slot * (kTileM * g.att_DH) + q * g.att_DH + dv);
            ml[pi][sp] = *reinterpret_cast<const f32x2*>(g.att_pml + slot * (kTileM * 2) + q * 2);
          }
        }
#pragma unroll
        for (int pi = 0; pi < PB; ++pi) {
          const int p = p0 + pi * nthr;
          const int row = p / ppr, c4 = p - row * ppr;
          float mstar = -1e30f;
#pragma unroll
          for (int sp = 0; sp < N; ++sp) mstar = fmaxf(mstar, sp < g.att_S ? ml[pi][sp][0] : -1e30f);
          f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
          float lsum = 0.f;
#pragma unroll
          for (int sp = 0; sp < N; ++sp) {
            const float w = (sp < g.att_S && ml[pi][sp][1] != 0.f) ? __expf(ml[pi][sp][0] - mstar) : 0.f;
            acc4 += po[pi][sp] * w;
            lsum += w * ml[pi][sp][1];
          }
          f32x4 v = acc4 * ((lsum != 0.f) ? 1.0f / lsum : 0.f);
          if (f.round16) v = r16_c<true>(v);
          if constexpr (F32) st4(smem + row * LDA + c4 * 4, v);
          else stage16<A16>(smem + row * LDA + (c4 >> 3) * 32, c4 & 7, v);
        }
      }
    };
    if (g.att_S <= 2) merge_pass(std::integral_constant<int, 2>{});
    else if (g.att_S == 3) merge_pass(std::integral_constant<int, 3>{});
    else if (g.att_S == 4) merge_pass(std::integral_constant<int, 4>{});
    else merge_pass(std::integral_constant<int, kMaxAttnSplits>{});
  }
  GC_RSTAMP(1);                                // ring issued; partials loaded, merged, split, staged
  __syncthreads();
  GC_RSTAMP(2);

  // ---- y^T = W^T x A^T over the whole K (transposed product: a lane gets 4 consecutive columns) ----
  f32x16 acc[MT][NT], accx[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        acc[mt][nt][q] = 0.f;
        accx[mt][nt][q] = 0.f;
      }
  {
    int s = 0;
    const float* arow = smem + r * LDA + hh * (F32 ? 8 : 4);
#pragma unroll 1
    for (int st = 0; st < steps; st += 8) {     // D % 128 == 0
      ws_quad<MT, NT, R, 0, A16, F32>(acc, accx, wh, wl, arow, 32 * LDA, st, wf, cts, s, steps);
      ws_quad<MT, NT, R, 4 % R, A16, F32>(acc, accx, wh, wl, arow, 32 * LDA, st + 4, wf, cts, s, steps);
    }
  }
  GC_RSTAMP(3);                                // products issued
  __syncthreads();                             // every wave is done with the A tile: it becomes y
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int cbase = (wave * NT + nt) * 32 + 4 * hh;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = hilo(acc[mt][nt][4 * j + e], accx[mt][nt][4 * j + e]);
        st4(smem + (mt * 32 + r) * LDA + cbase + 8 * j, v);
      }
    }
  __syncthreads();
  GC_RSTAMP(4);                                // y tile in LDS

  // ---- row pass (gc_rowop with one slab): wave w finishes rows w, w + nwave, ...; four rows per
  // pass, so that the four rows' loads of x are in flight together ----
  with_flag(f.round16, [&](auto rc) __attribute__((always_inline)) {
  constexpr bool RND = decltype(rc)::value;
  for (int rb = wave; rb < RH; rb += 4 * nwave) {
    f32x4 v[4][2];
    float s1[4], s2[4];
    if (kEarly && rb == wave) {                 // the first four rows were fetched at the top of the kernel
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 2; ++i) v[k][i] = xpre[k][i];
    } else {
      load_x_rows(rb, v);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int rr = rb + k * nwave;
      const int row = mtile * RH + rr;
      s1[k] = 0.f;
      s2[k] = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = 4 * lane + 256 * i;
        if (c < D && rr < RH) {
          f32x4 a = v[k][i];
          if constexpr (kEarly) a += bpre[i];
          else if (f.bias) a += ld4(f.bias + c);
          a += ld4(smem + rr * LDA + c);
          a = r16_c<RND>(a);
          if (row < g.rows) {
            if constexpr (A16) sth4(as_h16(f.x) + (size_t)row * D + c, a);
            else st4(f.x + (size_t)row * D + c, a);
          }
          v[k][i] = a;
          s1[k] += a[0] + a[1] + a[2] + a[3];
          s2[k] += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        s1[k] += __shfl_xor(s1[k], o);
        s2[k] += __shfl_xor(s2[k], o);
      }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int rr = rb + k * nwave;
      const int row = mtile * RH + rr;
      if (rr >= RH || row >= g.rows) continue;
      const float mean = s1[k] / (float)D;
      const float var = fmaxf(s2[k] / (float)D - mean * mean, 0.f);
      const float rstd = 1.0f / sqrtf(var + 1e-6f);
      const float* cs = f.cond + (size_t)(row % f.B) * f.cond_stride;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = 4 * lane + 256 * i;
        if (c < D) {
          f32x4 sc, of;
          if (kEarly && f.B == 1) {
            sc = scpre[i];
            of = ofpre[i];
          } else {                              // per-row batch element / 32-row form: fetched here (cache hits)
            sc = ld4(cs + c);
            of = ld4(cs + D + c);
          }
          if constexpr (A16) {
            sth4(as_h16(f.h) + (size_t)row * D + c, (v[k][i] - mean) * rstd * sc + of);
          } else {
            st4(f.h + (size_t)row * D + c, r16_c<RND>((v[k][i] - mean) * rstd * sc + of));
          }
        }
      }
    }
  }
  });
#ifdef GC_STAMPS
  GC_RSTAMP(5);                                // row pass: x / h stores issued
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  GC_RSTAMP(6);
  rst[7] = __builtin_amdgcn_s_memrealtime() - rrt0;
  if (g_rowop_stamps && lane == 0) {
    unsigned long long* o = g_rowop_stamps + ((size_t)blockIdx.x * nwave + wave) * 8;
    for (int i = 0; i < 8; ++i) o[i] = rst[i];
  }
#endif
}

template <int CLS>
static hipError_t launch_gemm_rowop_c(hipStream_t s, const GemmArgs& g, const RowFuse& f) {
  const int D = g.n;
  if (D % 128 || D > 512 || g.k_slice != D || g.ldw != D || !f.x || !f.h || !f.cond || f.B < 1 ||
      g.att_S > kMaxAttnSplits)
    return hipErrorInvalidValue;
  const int nt = D > 256 ? 2 : 1;
  const int nthr = (D / (32 * nt)) * 64;       // 256 (d = 128) or 512 threads
  // 16-row workgroups while 32-row ones would leave more than half of the 256 CUs idle; 48-row workgroups when the
  // 32-row tiles would need a partly filled extra round of workgroups and 48-row ones do not (see the kernel)
  static int rh_env = -1;
  if (rh_env < 0) {
    const char* e = getenv("GC_TUNE_OUT_RH");
    rh_env = (e && *e) ? atoi(e) : 0;
  }
  const int t32 = (g.rows + 31) / 32, t48 = (g.rows + 47) / 48;
  int rh = t32 <= 128 ? 16 : 32;
  if (t32 > 256 && (t32 + 255) / 256 > (t48 + 255) / 256 && nthr == 512 && g.att_S == 0) rh = 48;
  if (rh_env == 16 || rh_env == 32 || (rh_env == 48 && nthr == 512 && g.att_S == 0)) rh = rh_env;
  const size_t lds = (size_t)(rh > 32 ? 64 : 32) * (D + 4) * sizeof(float);
  const int grid = (g.rows + rh - 1) / rh;
  if (grid <= 0) return hipSuccess;
#define GC_ROWOP_R(NT_, AM_, RH_)                                                                          \
  {                                                                                                        \
    if constexpr (!kTuA16) {                                                                               \
      if (g.f32w) {                                                                                        \
        static DynLdsOnce once32;                                                                          \
        if (hipError_t e = once32.ensure((const void*)gc_gemm_rowop_kernel<NT_, AM_, CLS, RH_, false, true>, 64 * 516 * 4)) return e; \
        hipLaunchKernelGGL((gc_gemm_rowop_kernel<NT_, AM_, CLS, RH_, false, true>), dim3(grid), dim3(nthr), lds, s, g, f); \
        return hipGetLastError();                                                                          \
      }                                                                                                    \
    }                                                                                                      \
    static DynLdsOnce once;                                                                                \
    if (hipError_t e = once.ensure((const void*)gc_gemm_rowop_kernel<NT_, AM_, CLS, RH_, kTuA16>, 64 * 516 * 4)) return e; \
    hipLaunchKernelGGL((gc_gemm_rowop_kernel<NT_, AM_, CLS, RH_, kTuA16>), dim3(grid), dim3(nthr), lds, s, g, f); \
  }
#define GC_ROWOP(NT_, AM_) { if (rh == 16) GC_ROWOP_R(NT_, AM_, 16) else if (rh == 48 && AM_ == 0) GC_ROWOP_R(NT_, 0, 48) else GC_ROWOP_R(NT_, AM_, 32) }
  if (nt == 1 && g.att_S > 0) GC_ROWOP(1, 1)
  else if (nt == 1) GC_ROWOP(1, 0)
  else if (g.att_S > 0) GC_ROWOP(2, 1)
  else GC_ROWOP(2, 0)
#undef GC_ROWOP
#undef GC_ROWOP_R
  return hipGetLastError();
}

hipError_t launch_gemm_rowop(hipStream_t s, int cls, const GemmArgs& g, const RowFuse& f) {
  switch (cls) {
    case KC_GEMM_OUT: return launch_gemm_rowop_c<KC_GEMM_OUT>(s, g, f);
    default: return hipErrorInvalidValue;
  }
}

// ----------------------------------------------------------------------------
// gc_ffw_fused: FFW layer 1 + gelu + FFW layer 2 (sparse_transformer.py:252-268) in one launch.
// Workgroup = (32-row tile, hidden slice z of 256 columns), 4 waves:
//   phase 1  u[32 x 256] = gelu(a[32 x d] @ W1[:, slice] + b1)   wave w: hidden columns 64w .. 64w+63
//   phase 2  slab_z[32 x d] = u @ W2[slice, :]                    wave w: output columns (d/4)w ..
// Both weight matrices stream from their WF16 images straight into MFMA registers (as in
// gc_gemm_ws); a and u live in LDS as hi/lo halfs.  The 21-MB hidden activation round trip of
// the two-launch form (and one launch per layer) disappears; the price is f/256 slabs instead of 4
// for the following row pass.  blockIdx % 8 selects the hidden slice (f/256 == 8 at the GenCast
// sizes), so each XCD's L2 keeps exactly one 512-KB pair of weight slices.
// ----------------------------------------------------------------------------
// d = 128 * ND; 32 * MT rows per workgroup; NWC waves split the 256 hidden columns of the slice in
// phase 1 and the d output columns in phase 2 (4, or 8 with half the accumulators per wave)
template <int ND, int MT, int NWC = 4, bool A16 = false /* exact-fp16 a and hidden tile: 2 MFMAs per product */,
          bool F32 = false /* exact-f32 family: WF32 weight images, float32 tiles in LDS, v_mfma_f32_32x32x2_f32 */>
__global__ __launch_bounds__(64 * NWC, NWC == 8 ? 2 : ((ND <= 2 && MT == 1) ? 3 : (ND * MT <= 4 ? 2 : 1))) void gc_ffw_fused_kernel(FfwArgs g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int D = 128 * ND, LDA = D + 4, FS = 256, LDU = FS + 4, R = 4, BM = 32 * MT;
  constexpr int NTHR = 64 * NWC, NT1 = FS / (32 * NWC), NT2 = D / (32 * NWC);
  constexpr int AP = 32 * (D / 4) / NTHR;       // 16-byte pieces per thread per 32-row block of the a tile
  float* At = smem;                             // [BM][LDA]  S16
  float* Ut = smem;                             // [BM][LDU]  S16: takes over a's space after phase 1
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int S = g.f / FS;
  int z = blockIdx.x % S, mtile = blockIdx.x / S;
  if (g.xcd_tiles) {
    // (experiment, GC_TUNE_FFW_XCD=1) the S hidden slices of a row tile on ONE XCD: blocks with
    // equal blockIdx.x % 8 share an XCD, so XCD x takes row tiles x, x + 8, ...; its L2 then holds all S slabs of its
    // rows (the row pass is mapped the same way) and streams every weight slice instead of keeping one
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    mtile = (j / S) * 8 + xcd;
    z = j - (j / S) * S;
    if (mtile * BM >= g.rows) return;
  }
#ifdef GC_STAMPS
  unsigned long long stamp_v[8];
  int stamp_n = 0;
#define GC_STAMP() do { stamp_v[stamp_n++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GC_STAMP() do { } while (0)
#endif
  GC_STAMP();                                                    // 0: entry

  // phase-1 weight stream: column tiles (z*8 + wave*2 + nt) of W1^T, all d/16 steps
  constexpr int steps1 = D / 16;
  const float* wf1 = g.w1f + (size_t)(z * (FS / 32) + wave * NT1) * steps1 * 512 + lane * 4;
  f32x4 wh1[R][NT1], wl1[R][NT1];
  ws_ring_fill<NT1, R>(wh1, wl1, wf1, (size_t)steps1 * 512, steps1);

  if constexpr (A16) {   // a is stored as halfs (fp16 activation storage): 16-byte pieces of 8 k values, copied as they are
    static_assert(AP % 2 == 0, "fp16 piece mapping");
    f32x4 ra[MT][AP / 2];
#pragma unroll
    for (int mb = 0; mb < MT; ++mb)
#pragma unroll
      for (int i = 0; i < AP / 2; ++i) {
        const int p = tid + NTHR * i, row = mb * 32 + p / (D / 8), c8 = p % (D / 8);
        int grow = mtile * BM + row;
        if (grow >= g.rows) grow = g.rows - 1;
        ra[mb][i] = ld4(reinterpret_cast<const float*>(as_h16(g.a) + (size_t)grow * D + c8 * 8));
      }
#pragma unroll
    for (int mb = 0; mb < MT; ++mb)
#pragma unroll
      for (int i = 0; i < AP / 2; ++i) {
        const int p = tid + NTHR * i, row = mb * 32 + p / (D / 8), c8 = p % (D / 8);
        st4(At + row * LDA + (c8 >> 2) * 32 + (c8 & 3) * 4, ra[mb][i]);
      }
  } else {   // a tile -> LDS (hi/lo): every piece of the 32*MT rows is requested before the first is staged
    f32x4 ra[MT][AP];
#pragma unroll
    for (int mb = 0; mb < MT; ++mb)
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        const int p = tid + NTHR * i, row = mb * 32 + p / (D / 4), c4 = p % (D / 4);
        int grow = mtile * BM + row;
        if (grow >= g.rows) grow = g.rows - 1;
        ra[mb][i] = ld4(g.a + (size_t)grow * D + c4 * 4);
      }
#pragma unroll
    for (int mb = 0; mb < MT; ++mb)
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        const int p = tid + NTHR * i, row = mb * 32 + p / (D / 4), c4 = p % (D / 4);
        if constexpr (F32) st4(At + row * LDA + c4 * 4, ra[mb][i]);
        else stage16<A16>(At + row * LDA + (c4 >> 3) * 32, c4 & 7, ra[mb][i]);
      }
  }
  GC_STAMP();                                                    // 1: a tile loaded, split and written to LDS
  __syncthreads();
  GC_STAMP();                                                    // 2: barrier passed

  // layer-1 bias of this lane's hidden columns: needed after the first product, requested before it (8-wave
  // form: 4 registers; the 4-wave forms have no registers to spare and fetch it where it is used)
  constexpr bool kB1Early = NWC == 8;
  f32x4 b1pre[kB1Early ? NT1 : 1][4];
  if constexpr (kB1Early) {
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) b1pre[nt][j] = ld4(g.b1 + z * FS + (wave * NT1 + nt) * 32 + 4 * hh + 8 * j);
  }

  f32x16 acc1[MT][NT1], accx1[MT][NT1];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        acc1[mt][nt][q] = 0.f;
        accx1[mt][nt][q] = 0.f;
      }
  {
    int s = 0;
    const float* arow = At + r * LDA + hh * (F32 ? 8 : 4);
#pragma unroll 1
    for (int st = 0; st < steps1; st += 4)
      ws_quad<MT, NT1, R, 0, A16, F32>(acc1, accx1, wh1, wl1, arow, 32 * LDA, st, wf1, (size_t)steps1 * 512, s, steps1);
  }
  GC_STAMP();                                                    // 3: phase-1 products issued
  // phase-2 weight stream: column tiles (wave*ND + nt) of W2^T, steps z*16 .. z*16+15 of f/16
  const int steps2_total = g.f / 16;
  const float* wf2 = g.w2f + ((size_t)(wave * NT2) * steps2_total + (size_t)z * (FS / 16)) * 512 + lane * 4;
  const size_t cts2 = (size_t)steps2_total * 512;
  f32x4 wh2[R][NT2], wl2[R][NT2];
  ws_ring_fill<NT2, R>(wh2, wl2, wf2, cts2, FS / 16);   // in flight while the hidden tile is finished

  // u = gelu(acc + b1), split to hi / lo halfs IN REGISTERS first, then the barrier: a wave that finishes its
  // products early does this vector work while slower waves are still on the matrix pipe.  (With the barrier
  // in front all eight waves did it in lockstep with the matrix pipe idle -- 28 % of a wave's lifetime in the
  // in-kernel stamps of tools/stamp_ffw.cpp.)  A lane holds 4 consecutive hidden columns of row r.
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  typedef float f32x2v __attribute__((ext_vector_type(2)));
  f16x4 uh[NT1][4][MT], ul[NT1][4][MT];
  with_flag(g.round16, [&](auto rc) __attribute__((always_inline)) {
    constexpr bool RND = decltype(rc)::value;
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt) {
      const int cbase = (wave * NT1 + nt) * 32 + 4 * hh;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 bv;
        if constexpr (kB1Early) bv = b1pre[nt][j];
        else bv = ld4(g.b1 + z * FS + cbase + 8 * j);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          _Float16 hv[4], lv[4];
          if constexpr (F32) {                    // float32 hidden tile: the same 8 bytes per pair of registers, unsplit
            f32x4 uf;
#pragma unroll
            for (int e = 0; e < 4; ++e) uf[e] = r16_c<RND>(gelu_tanh_fast(acc1[mt][nt][4 * j + e] + bv[e]));
            uh[nt][j][mt] = __builtin_bit_cast(f16x4, f32x2v{uf[0], uf[1]});
            ul[nt][j][mt] = __builtin_bit_cast(f16x4, f32x2v{uf[2], uf[3]});
            continue;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float u = r16_c<RND>(gelu_tanh_fast(hilo(acc1[mt][nt][4 * j + e], accx1[mt][nt][4 * j + e]) + bv[e]));
            if constexpr (A16) {                  // exact fp16 already: the lo plane is zero and never read
              hv[e] = (_Float16)u;
              lv[e] = (_Float16)0.f;
            } else {
              split16(u, hv[e], lv[e]);
            }
          }
          uh[nt][j][mt] = f16x4{hv[0], hv[1], hv[2], hv[3]};
          ul[nt][j][mt] = f16x4{lv[0], lv[1], lv[2], lv[3]};
        }
      }
    }
  });
  __syncthreads();                              // every wave has finished reading the a tile
#pragma unroll
  for (int nt = 0; nt < NT1; ++nt) {
    const int cbase = (wave * NT1 + nt) * 32 + 4 * hh;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int col = cbase + 8 * j;
        if constexpr (F32) {
          float* pf = Ut + (size_t)(mt * 32 + r) * LDU + col;
          *reinterpret_cast<f16x4*>(pf) = uh[nt][j][mt];
          *reinterpret_cast<f16x4*>(pf + 2) = ul[nt][j][mt];
          continue;
        }
        _Float16* p = reinterpret_cast<_Float16*>(Ut + (size_t)(mt * 32 + r) * LDU + (col & ~31)) + (col & 31);
        *reinterpret_cast<f16x4*>(p) = uh[nt][j][mt];
        if constexpr (!A16) *reinterpret_cast<f16x4*>(p + 32) = ul[nt][j][mt];
      }
  }
  __syncthreads();
  GC_STAMP();                                                    // 4: gelu + hidden tile in LDS + barrier

  f32x16 acc2[MT][NT2], accx2[MT][NT2];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        acc2[mt][nt][q] = 0.f;
        accx2[mt][nt][q] = 0.f;
      }
  {
    int s = 0;
    const float* urow = Ut + r * LDU + hh * (F32 ? 8 : 4);
#pragma unroll 1
    for (int st = 0; st < FS / 16; st += 4)
      ws_quad<MT, NT2, R, 0, A16, F32>(acc2, accx2, wh2, wl2, urow, 32 * LDU, st, wf2, cts2, s, FS / 16);
  }
  GC_STAMP();                                                    // 5: phase-2 products issued
  // slab z: lane (r, hh) owns 4 consecutive columns of row r in every 8-column group
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int grow = mtile * BM + mt * 32 + r;
    if (grow >= g.rows) continue;
    float* orow = g.out + ((size_t)z * g.rows + grow) * D;
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) {
      const int cbase = (wave * NT2 + nt) * 32 + 4 * hh;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = hilo(acc2[mt][nt][4 * j + e], accx2[mt][nt][4 * j + e]);
        if (g.wt) st4_wt(orow + cbase + 8 * j, v);
        else st4(orow + cbase + 8 * j, v);
      }
    }
  }
#ifdef GC_STAMPS
  GC_STAMP();                                                    // 6: slab stores issued
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  GC_STAMP();                                                    // 7: slab stores drained
  if (g.stamps && lane == 0)
    for (int i = 0; i < 8; ++i) g.stamps[((size_t)blockIdx.x * NWC + wave) * 8 + i] = stamp_v[i];
#endif
#undef GC_STAMP
}

template <int ND, int MT, int NWC = 4>
static hipError_t launch_ffw_fused_t(hipStream_t s, const FfwArgs& g) {
  const size_t lds = (size_t)(32 * MT * ((ND > 2 ? 128 * ND : 256) + 4)) * sizeof(float);
  const int n_mt = (g.rows + 32 * MT - 1) / (32 * MT);
  const int grid = g.xcd_tiles ? 8 * ((n_mt + 7) / 8) * (g.f / 256) : n_mt * (g.f / 256);
  if (grid <= 0) return hipSuccess;
  if constexpr (!kTuA16) {
    if (g.f32w) {                                // exact-f32 family: WF32 images
      static DynLdsOnce once32;
      if (hipError_t e = once32.ensure((const void*)gc_ffw_fused_kernel<ND, MT, NWC, false, true>, (int)lds)) return e;
      hipLaunchKernelGGL((gc_ffw_fused_kernel<ND, MT, NWC, false, true>), dim3(grid), dim3(64 * NWC), lds, s, g);
      return hipGetLastError();
    }
  }
  static DynLdsOnce once;
  if (hipError_t e = once.ensure((const void*)gc_ffw_fused_kernel<ND, MT, NWC, kTuA16>, (int)lds)) return e;
  hipLaunchKernelGGL((gc_ffw_fused_kernel<ND, MT, NWC, kTuA16>), dim3(grid), dim3(64 * NWC), lds, s, g);
  return hipGetLastError();
}

hipError_t launch_ffw_fused(hipStream_t s, const FfwArgs& g) {
  if (g.d % 128 || g.d > 512 || g.f % 256 || g.f < 256) return hipErrorInvalidValue;
  // 64-row tiles halve the weight traffic (the kernel streams both slices per tile and is bound by
  // the L2 -> CU rate); used once they still give more than one workgroup per CU
  static int mt2 = -1;
  if (mt2 < 0) {
    const char* e = getenv("GC_TUNE_FFW_MT");
    mt2 = e ? atoi(e) : 0;
  }
  const bool big = mt2 == 2 || (mt2 == 0 && ((g.rows + 63) / 64) * (g.f / 256) >= 300);
  switch (g.d / 128) {
    case 1: return big ? launch_ffw_fused_t<1, 2>(s, g) : launch_ffw_fused_t<1, 1>(s, g);
    case 2: {
      // 96-row tiles on 8 waves when that gives one balanced wave of workgroups (between half a chip and
      // a chip: 27 x 8 = 216 at the nano size).  With 64-row tiles 328 workgroups land two on some CUs and
      // one on the others, and the kernel lasts as long as the doubly loaded CUs (32.2 vs 29.2 us).
      const int b96 = ((g.rows + 95) / 96) * (g.f / 256);
      if (mt2 == 38 || (mt2 == 0 && b96 > 128 && b96 <= 256)) return launch_ffw_fused_t<2, 3, 8>(s, g);
      if (mt2 == 28) return launch_ffw_fused_t<2, 2, 8>(s, g);
      return big ? launch_ffw_fused_t<2, 2>(s, g) : launch_ffw_fused_t<2, 1>(s, g);
    }
    case 4: return launch_ffw_fused_t<4, 1>(s, g);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_gemm_ws(hipStream_t s, int cls, const GemmArgs& g, int mt, int splits, int epi) {
  switch (cls) {
    case KC_GEMM_QKV: return launch_gemm_ws_c<KC_GEMM_QKV>(s, g, mt, splits, epi);
    case KC_GEMM_OUT: return launch_gemm_ws_c<KC_GEMM_OUT>(s, g, mt, splits, epi);
    case KC_GEMM_FFW1: return launch_gemm_ws_c<KC_GEMM_FFW1>(s, g, mt, splits, epi);
    case KC_GEMM_FFW2: return launch_gemm_ws_c<KC_GEMM_FFW2>(s, g, mt, splits, epi);
    case KC_GEMM_NODE: return launch_gemm_ws_c<KC_GEMM_NODE>(s, g, mt, splits, epi);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_gemm(hipStream_t s, int cls, const GemmArgs& g, int shape, int splits, int epi, bool f16) {
  switch (cls) {
    case KC_GEMM_QKV: return launch_gemm_c<KC_GEMM_QKV>(s, g, shape, splits, epi, f16);
    case KC_GEMM_OUT: return launch_gemm_c<KC_GEMM_OUT>(s, g, shape, splits, epi, f16);
    case KC_GEMM_FFW1: return launch_gemm_c<KC_GEMM_FFW1>(s, g, shape, splits, epi, f16);
    case KC_GEMM_FFW2: return launch_gemm_c<KC_GEMM_FFW2>(s, g, shape, splits, epi, f16);
    case KC_GEMM_NODE: return launch_gemm_c<KC_GEMM_NODE>(s, g, shape, splits, epi, f16);
    default: return hipErrorInvalidValue;
  }
}

// ----------------------------------------------------------------------------
// gc_rowop: the residual stream's row pass.  One wave per row:
//   x <- x + bias + sum_s partial[s]          (split-K slabs summed in slab order)
//   h <- cond(LayerNorm(x))                   (input of the next projection)
// Residual adds + pre-norms of Block.__call__ (sparse_transformer.py:518-524) and
// the final norm (:630-633).
// ----------------------------------------------------------------------------
// H16: x and h are _Float16 arrays (physical fp16 activation storage, implies round16); slabs, bias, conditioning float32.
template <int NS /* slabs, compile time (loads unconditional); -1: any number, fetched one by one */, bool H16 = false>
__global__ __launch_bounds__(256) void gc_rowop_kernel(float* __restrict__ x,
                                                        const float* __restrict__ bias,
                                                        const float* __restrict__ partials, int n_slabs,
                                                        int rows, int d, int B,
                                                        const float* __restrict__ cond, int cond_stride,
                                                        float* __restrict__ h, int h_s16, int round16, int xcd_tile_rows) {
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (xcd_tile_rows) {                          // (GC_TUNE_FFW_XCD=1) rows of fused-FFW row tile t on the XCD that wrote its slabs: t % 8
    const int bpt = xcd_tile_rows >> 2, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    row = ((j / bpt) * 8 + xcd) * xcd_tile_rows + (j % bpt) * 4 + (threadIdx.x >> 6);
  }
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const size_t slab = (size_t)rows * d;
  // Every load of the row is issued before the first one is used: x, bias, the conditioning vectors and the NS
  // partial slabs (written as `for (slab) a += load` with a run-time count hipcc emits one load + s_waitcnt
  // vmcnt(0) per slab: ten dependent memory round trips for a row, most of this kernel's 7.3 us at 8 slabs).
  // The slabs are still added in slab order, so the sums keep their bits.
  const float* cs = cond + (size_t)(row % B) * cond_stride;
  float4 v[2], scv[2], ofv[2];
  float s1 = 0.f, s2 = 0.f;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = 4 * lane + 256 * i;
    v[i] = zero4; scv[i] = zero4; ofv[i] = zero4;
    if (c < d) {
      const float* pr = partials + (size_t)row * d + c;
      float4 p[NS > 0 ? NS : 1];
      if constexpr (NS > 0) {                  // the unconditional loads go first: hipcc waits for pending loads
#pragma unroll                                 // at the joins of the `if (bias)` / `if (h)` branches below
        for (int j = 0; j < NS; ++j) p[j] = *reinterpret_cast<const float4*>(pr + (size_t)j * slab);
      }
      float4 a;
      if constexpr (H16) {
        const f32x4 xa = ldh4(as_h16(x) + (size_t)row * d + c);
        a = make_float4(xa[0], xa[1], xa[2], xa[3]);
      } else {
        a = *reinterpret_cast<const float4*>(x + (size_t)row * d + c);
      }
      float4 bb = zero4;
      if (bias) bb = *reinterpret_cast<const float4*>(bias + c);
      if (h) {
        scv[i] = *reinterpret_cast<const float4*>(cs + c);
        ofv[i] = *reinterpret_cast<const float4*>(cs + d + c);
      }
      if constexpr (NS > 0) {
        a.x += bb.x; a.y += bb.y; a.z += bb.z; a.w += bb.w;
#pragma unroll
        for (int j = 0; j < NS; ++j) { a.x += p[j].x; a.y += p[j].y; a.z += p[j].z; a.w += p[j].w; }
      } else {
        a.x += bb.x; a.y += bb.y; a.z += bb.z; a.w += bb.w;
        for (int sidx = 0; sidx < n_slabs; ++sidx) {
          const float4 q = *reinterpret_cast<const float4*>(pr + (size_t)sidx * slab);
          a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w;
        }
      }
      if (round16) { a.x = r16(a.x); a.y = r16(a.y); a.z = r16(a.z); a.w = r16(a.w); }
      if (n_slabs > 0 || bias) {
        if constexpr (H16) sth4(as_h16(x) + (size_t)row * d + c, f32x4{a.x, a.y, a.z, a.w});
        else *reinterpret_cast<float4*>(x + (size_t)row * d + c) = a;
      }
      v[i] = a;
      s1 += a.x + a.y + a.z + a.w;
      s2 += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
    }
  }
  if (!h) return;
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  const float mean = s1 / (float)d;
  const float var = fmaxf(s2 / (float)d - mean * mean, 0.f);
  const float rstd = 1.0f / sqrtf(var + 1e-6f);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = 4 * lane + 256 * i;
    if (c < d) {
      const float4 sc = scv[i];
      const float4 of = ofv[i];
      float4 o;
      o.x = (v[i].x - mean) * rstd * sc.x + of.x;
      o.y = (v[i].y - mean) * rstd * sc.y + of.y;
      o.z = (v[i].z - mean) * rstd * sc.z + of.z;
      o.w = (v[i].w - mean) * rstd * sc.w + of.w;
      if (round16) { o.x = r16(o.x); o.y = r16(o.y); o.z = r16(o.z); o.w = r16(o.w); }
      if constexpr (H16) sth4(as_h16(h) + (size_t)row * d + c, f32x4{o.x, o.y, o.z, o.w});
      else if (h_s16) store4_s16(h, (size_t)row, d, c, o.x, o.y, o.z, o.w);
      else *reinterpret_cast<float4*>(h + (size_t)row * d + c) = o;
    }
  }
}

hipError_t launch_rowop(hipStream_t s, float* x, const float* bias, const float* partials, int n_slabs,
                        int rows, int d, int B, const float* cond, int cond_stride, float* h, int h_s16,
                        bool round16, bool h16, int xcd_tile_rows) {
  if (d > 512 || d % 4 || h_s16 < 0 || h_s16 > 1 || (h_s16 && d % 32) || (h16 && h_s16 == 1)) return hipErrorInvalidValue;
  if (xcd_tile_rows % 4) return hipErrorInvalidValue;
  const int n_blocks = xcd_tile_rows ? 8 * (((rows + xcd_tile_rows - 1) / xcd_tile_rows + 7) / 8) * (xcd_tile_rows / 4) : (rows + 3) / 4;
#define GC_ROWOP_NS(NS_)                                                                                   \
  if (h16) hipLaunchKernelGGL((gc_rowop_kernel<NS_, true>), dim3(n_blocks), dim3(256), 0, s, x, bias, partials, n_slabs, \
                     rows, d, B, cond, cond_stride, h, h_s16, 1, xcd_tile_rows);                           \
  else hipLaunchKernelGGL((gc_rowop_kernel<NS_, false>), dim3(n_blocks), dim3(256), 0, s, x, bias, partials, n_slabs, \
                     rows, d, B, cond, cond_stride, h, h_s16, round16 ? 1 : 0, xcd_tile_rows)
  switch (n_slabs) {                             // the counts the forward pass uses; anything else: generic
    case 1: GC_ROWOP_NS(1); break;
    case 2: GC_ROWOP_NS(2); break;
    case 4: GC_ROWOP_NS(4); break;
    case 8: GC_ROWOP_NS(8); break;
    default: GC_ROWOP_NS(-1); break;
  }
#undef GC_ROWOP_NS
  return hipGetLastError();
}

// ----------------------------------------------------------------------------
// gc_attention: softmax over each mesh node's k-hop neighbourhood.
// Reference: TriblockdiagMHA (sparse_transformer.py:309-349) computes dense
// 3-block-wide logits and masks them to -1e30; masked terms contribute exactly
// 0, so the function is the per-node neighbourhood softmax.
//
// Work decomposition: mesh nodes are renumbered (gc_graph.cpp) so that 32
// consecutive nodes form a spatially compact query tile.  One workgroup = one
// tile, one wave per head.  For the tile the host precomputed the sorted UNION
// of its queries' neighbourhoods (padded to 32-key chunks) and a 32-bit
// membership mask per (chunk, query).  Per chunk a wave computes
//   S^T[key][q] = K.Q^T           (MFMA, A = K rows gathered by index)
//   p = mask ? exp(s - m) : 0      (online softmax, lazy rescale)
//   O[q][dv]  += P.V               (MFMA; S^T's accumulator IS the A operand)
// K/V rows are read straight from L2 into MFMA operand registers; each gathered
// row is reused by 32 queries.
// ----------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(DH >= 128 ? 256 : 512) void gc_attention_kernel(
    const float* __restrict__ qkv, float* __restrict__ o, float* __restrict__ part_o,
    float* __restrict__ part_ml, int M, int B, int D, int S, int out_s16,
    const int* __restrict__ tile_chunk_start, const int* __restrict__ union_idx,
    const unsigned* __restrict__ mask_bits, int feat16, const int* __restrict__ items, int n_items) {
  constexpr int HK = DH / 2;   // k-steps of the QK^T product (2 per MFMA across lane halves)
  constexpr int NS = DH / 32;  // 32-wide dv slices
  // items != nullptr: the grid runs a host-made work-item list (gc_api.hip build_attention_items; see gc_attention_v2_kernel)
  // instead of (tile, split) pairs -- one whole tile per CU, then the remaining tiles as key-range pieces whose partials
  // the out-projection's loader merges.  Consecutive items go to one XCD (blockIdx % 8), as in the v2 kernel.
  int t = blockIdx.x, sp = blockIdx.y, it_lo = 0, it_hi = 0, it_slot = -1;
  const int b = blockIdx.z;
  if (items) {
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int lin = xcd * (n_items >> 3) + jj;       // n_items % 8 == 0
    t = items[4 * lin];
    if (t < 0) return;
    it_lo = items[4 * lin + 1]; it_hi = items[4 * lin + 2]; it_slot = items[4 * lin + 3];
    sp = 0;
  }
  const int head = threadIdx.x >> 6, H = blockDim.x >> 6;
  const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
  const size_t ld = (size_t)3 * D;
  const float scale = 1.0f / sqrtf((float)DH);
  const float kNegBig = -1e30f;
  const float kThr = feat16 ? 10.0f : 40.0f;  // lazy-rescale threshold: p <= e^40 is far inside f32 range (e^10 inside fp16)

  int qnode = t * kTileM + r;
  if (qnode >= M) qnode = M - 1;
  float qf[HK];
  {
    const float* qp = qkv + ((size_t)qnode * B + b) * ld + head * DH + hh * HK;
#pragma unroll
    for (int i = 0; i < HK; i += 4) {
      const float4 v = *reinterpret_cast<const float4*>(qp + i);
      qf[i] = v.x * scale;
      qf[i + 1] = v.y * scale;
      qf[i + 2] = v.z * scale;
      qf[i + 3] = v.w * scale;
    }
  }
  f32x16 oacc[NS];
#pragma unroll
  for (int sl = 0; sl < NS; ++sl)
#pragma unroll
    for (int g = 0; g < 16; ++g) oacc[sl][g] = 0.f;
  float m_run = kNegBig, l_run = 0.f;

  // this block's share of the tile's key chunks
  int lo, hi;
  if (items) {
    lo = it_lo;
    hi = it_hi;
  } else {
    const int c_begin = tile_chunk_start[t], nc = tile_chunk_start[t + 1] - c_begin;
    lo = c_begin + (nc * sp) / S;
    hi = c_begin + (nc * (sp + 1)) / S;
  }
  const float* kbase = qkv + (size_t)b * ld + D + head * DH + hh * HK;
  const float* vbase = qkv + (size_t)b * ld + 2 * D + head * DH + r;

  float kf[HK];
  if (lo < hi) {
    const float* kp = kbase + (size_t)union_idx[lo * 32 + r] * B * ld;
#pragma unroll
    for (int i = 0; i < HK; i += 4) {
      const float4 v = *reinterpret_cast<const float4*>(kp + i);
      kf[i] = v.x; kf[i + 1] = v.y; kf[i + 2] = v.z; kf[i + 3] = v.w;
    }
  }
  for (int c = lo; c < hi; ++c) {
    // ---- issue this chunk's V loads and the next chunk's K loads first ------------
    // V row of accumulator register g on lane-half hh: key(g, hh) = (g&3) + 8*(g>>2) + 4*hh
    float vv[16][NS];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int4 vi = *reinterpret_cast<const int4*>(union_idx + c * 32 + 8 * q4 + 4 * hh);
      const int vidx[4] = {vi.x, vi.y, vi.z, vi.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* vp = vbase + (size_t)vidx[e] * B * ld;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) vv[4 * q4 + e][sl] = vp[sl * 32];
      }
    }
    float kn[HK];
    {
      const int cn = (c + 1 < hi) ? c + 1 : c;
      const float* kp = kbase + (size_t)union_idx[cn * 32 + r] * B * ld;
#pragma unroll
      for (int i = 0; i < HK; i += 4) {
        const float4 v = *reinterpret_cast<const float4*>(kp + i);
        kn[i] = v.x; kn[i + 1] = v.y; kn[i + 2] = v.z; kn[i + 3] = v.w;
      }
    }
    const unsigned mb = mask_bits[c * 32 + r];

    // ---- S^T = K . Q^T ------------------------------------------------------------
    f32x16 st;
#pragma unroll
    for (int g = 0; g < 16; ++g) st[g] = 0.f;
#pragma unroll
    for (int j = 0; j < HK; ++j) st = mfma32(kf[j], qf[j], st);

    // ---- masked online softmax (query = this lane's r; keys in registers) -----------
    float cmax = kNegBig;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const bool on = (mb >> acc_row(g, hh)) & 1u;
      cmax = on ? fmaxf(cmax, st[g]) : cmax;
    }
    cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
    const bool need = cmax > m_run + kThr;
    if (__any(need)) {
      const float m_new = need ? cmax : m_run;
      const float alpha = __expf(m_run - m_new);  // 1 where nothing changed, 0 on first use
      l_run *= alpha;
      m_run = m_new;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const float af = __shfl(alpha, acc_row(g, hh));
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) oacc[sl][g] *= af;
      }
    }
    float psum = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const bool on = (mb >> acc_row(g, hh)) & 1u;
      const float p = r16_if(on ? __expf(st[g] - m_run) : 0.f, feat16);   // fp16-feature mode: softmax weights are fp16
      st[g] = p;
      psum += p;
    }
    psum += __shfl_xor(psum, 32);
    l_run += psum;

    // ---- O += P . V  (S^T's accumulator is the A operand) ------------------------------
#pragma unroll
    for (int g = 0; g < 16; ++g)
#pragma unroll
      for (int sl = 0; sl < NS; ++sl) oacc[sl] = mfma32(st[g], vv[g][sl], oacc[sl]);
#pragma unroll
    for (int i = 0; i < HK; ++i) kf[i] = kn[i];
  }

  if (items ? (it_slot < 0) : (S == 1)) {
    const float inv_l = (l_run != 0.f) ? 1.0f / l_run : 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int qrow = acc_row(g, hh);
      const float il = __shfl(inv_l, qrow);
      const int node = t * kTileM + qrow;
      if (node < M) {
        const size_t orow = (size_t)node * B + b;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
          if (out_s16) store_s16(o, orow, D, head * DH + sl * 32 + r, oacc[sl][g] * il);
          else o[orow * D + head * DH + sl * 32 + r] = r16_if(oacc[sl][g] * il, feat16);
        }
      }
    }
  } else {
    const size_t slot = ((items ? (size_t)it_slot : (size_t)t * S + sp) * B + b) * H + head;
    float* po = part_o + slot * (kTileM * DH);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int qrow = acc_row(g, hh);
#pragma unroll
      for (int sl = 0; sl < NS; ++sl) po[qrow * DH + sl * 32 + r] = oacc[sl][g];
    }
    if (hh == 0) {
      float* pm = part_ml + slot * (kTileM * 2);
      pm[r * 2] = m_run;
      pm[r * 2 + 1] = l_run;
    }
  }
}

// f16x3 form of gc_attention_kernel (same decomposition, same operand roles): both products
// run as 3 fp16 MFMAs on hi/lo-split operands.  The f32 MFMAs were ~65 % of a chunk's time (64
// v_mfma_f32_32x32x2_f32 of 64 cycles each per 32-key chunk against 24 fp16 MFMAs of 32);
// K, V and P are split in registers right after they arrive.
// k-order: the 32x32x16 MFMA only needs A and B to agree on which k sits in which slot, so
//   QK^T step s: lane half hh contributes d = hh*DH/2 + 8s .. +7 (the contiguous half-row it loads);
//   P.V  step u: lane half hh contributes the 8 keys of accumulator registers 8u .. 8u+7, i.e.
//                exactly the layout S^T's accumulator already has.
__device__ __forceinline__ void split8(const float* x, f32x4& hi, f32x4& lo) {
#pragma clang fp contract(off)   // as split16: the residual of the rounded value, whatever expression produced it
  f16x8 h, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = x[e];       // no clamp: see split16 (Q, K, V arrive range-checked from the QKV projection)
    const _Float16 hv = (_Float16)v;
    h[e] = hv;
    l[e] = (_Float16)((v - (float)hv) * kLoScale);
  }
  hi = __builtin_bit_cast(f32x4, h);
  lo = __builtin_bit_cast(f32x4, l);
}

// hi half only: for values that are exactly fp16 already (fp16-feature mode)
__device__ __forceinline__ void split8_hi(const float* x, f32x4& hi) {
  f16x8 h;
#pragma unroll
  for (int e = 0; e < 8; ++e) h[e] = (_Float16)x[e];
  hi = __builtin_bit_cast(f32x4, h);
}

// ----------------------------------------------------------------------------
// gc_attention_v2: one workgroup = one 32-query tile x one key split (or one item of a work-item list), one wave per
// head, online softmax over 32-key chunks of the tile's key union, fed from the fp16 hi / lo planes the QKV projection
// already wrote (launch_gemm_ws epi 3):
//   * K fragments are loaded straight into MFMA operand registers (16-byte loads of the planes): no
//     per-tile re-splitting of every gathered key (it was ~190 of a chunk's ~900 vector instructions in round 1's form);
//   * V rows are gathered with 16-byte loads, staged per wave in LDS as they are ([key][dv], fp16), and
//     read back as the P.V B operand with ds_read_b64_tr_b16, the hardware transposed read (a lane
//     needs 8 KEYS of one dv column): 8 + 8 wide loads and 16 transposed reads per chunk replace 32
//     scalar gathers and another ~190 split instructions;  the 16-byte chunks of a row are XOR-swizzled
//     so that the four rows one transposed read touches fall into different bank quarters.
// ----------------------------------------------------------------------------
typedef __fp16 h4raw __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 join_tr(h4raw a, h4raw b) {
  u64x2 v = {__builtin_bit_cast(unsigned long long, a), __builtin_bit_cast(unsigned long long, b)};
  return __builtin_bit_cast(f32x4, v);
}

#ifdef GC_STAMPS
// Diagnostic builds (tools/stamp_attention.py): per-wave s_memtime stamps of the attention kernel.
__device__ unsigned long long* g_att_stamps = nullptr;
hipError_t set_attention_stamp_buffer(unsigned long long* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_att_stamps), &p, sizeof(p));
}
#define GC_ASTAMP(i) do { __builtin_amdgcn_sched_barrier(0); ast[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define GC_ASTAMP_ACC(i, t0) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t__ = __builtin_amdgcn_s_memtime(); ast[i] += t__ - (t0); (t0) = t__; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define GC_ASTAMP(i) do { } while (0)
#define GC_ASTAMP_ACC(i, t0) do { } while (0)
#endif

template <int DH>
__device__ __forceinline__ int v2_swz(int row) {           // XOR applied to a row's 16-byte chunk index
  return DH == 64 ? 4 * ((row >> 1) & 1) : (DH == 128 ? 4 * (row & 3) : 0);
}

// H16 (with FEAT16 only: physical fp16 activation storage): q is a _Float16 array [M * B][D] whose 16-byte pieces
// ARE the MFMA fragments (no split, no conversion), and the S == 1 output o is a _Float16 array.
// ITEMS (a launch whose tiles, one per workgroup and CU, would be a full round of workgroups plus a partly filled one:
// 321 tiles on 256 CUs at the 1-degree size): the grid runs a host-made list of work items instead of (tile, split)
// pairs -- `items[lin] = (tile, first chunk, end chunk, partial slot or -1)`.  One whole tile per CU first (slot -1: the
// output row is final), then the remaining tiles cut into key-range pieces, one per CU, whose partial (m, l, O) triples
// the out-projection's loader merges for those rows only (gc_gemm_rowop, att_tiles).  tile < 0: padding, nothing to do.
template <int DH, bool FEAT16, bool H16 = false, bool ITEMS = false>
__global__ __launch_bounds__(DH >= 128 ? 256 : 512) void gc_attention_v2_kernel(
    const float* __restrict__ qkv, const _Float16* __restrict__ kv16, float* __restrict__ o,
    float* __restrict__ part_o, float* __restrict__ part_ml, int M, int B, int D, int S,
    const int* __restrict__ tile_chunk_start, const int* __restrict__ union_idx,
    const unsigned* __restrict__ mask_bits, int n_tiles, int max_chunks, const int* __restrict__ items) {
  static_assert(!H16 || FEAT16, "fp16 storage implies fp16 features");
  constexpr int HK = DH / 2;       // q / k values of one row held by one lane half
  constexpr int KS = DH / 16;      // k16 steps of the QK^T product
  constexpr int NS = DH / 32;      // 32-wide dv slices
  constexpr int NP = FEAT16 ? 1 : 2;   // planes read: hi (+ lo)
  constexpr int CPR = DH / 8;      // 16-byte chunks per V row
  constexpr int NPV = DH / 16;     // V pieces per lane per plane per chunk (32 * CPR / 64)
  constexpr bool QL = DH >= 64;   // q fragments parked in LDS (lane-private slots) instead of 16 * NP registers
  const int n_pairs = ITEMS ? n_tiles /* = the number of items, a multiple of 8 */ : n_tiles * S;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int base_cnt = n_pairs >> 3, extra = n_pairs & 7;
  if (jj >= base_cnt + (xcd < extra ? 1 : 0)) return;
  const int lin = xcd * base_cnt + (xcd < extra ? xcd : extra) + jj;
#if defined(GC_EXP_ATT_256)      // timing-only ablation: one round of workgroups (the first 32 pairs of every XCD)
  if (jj >= 32) return;
#endif
  int t, sp, it_lo = 0, it_hi = 0, it_slot = -1;
  if constexpr (ITEMS) {
    t = items[4 * lin];
    if (t < 0) return;
    it_lo = items[4 * lin + 1]; it_hi = items[4 * lin + 2]; it_slot = items[4 * lin + 3];
    sp = 0;
  } else {
    t = lin / S;
    sp = lin - t * S;
  }
  const int b = blockIdx.z;
  const int head = threadIdx.x >> 6, H = blockDim.x >> 6;
  const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
  const float scale = 1.0f / sqrtf((float)DH);
  const float kNegBig = -1e30f;
  const float kThr = 10.0f;        // lazy-rescale threshold: p <= e^10 stays inside fp16 range
#ifdef GC_STAMPS
  unsigned long long ast[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tph = 0;
  const unsigned long long art0 = __builtin_amdgcn_s_memrealtime();   // constant 100 MHz
#endif
  GC_ASTAMP(0);

  // ---- Q: f32 from the projection output.  Only the LOADS are issued here; the split (~250 vector
  // instructions) is done further down, behind the first K / V gathers, whose latency it then covers (the
  // in-kernel stamps showed the prologue as three serial waits: q, the index staging, the first K / V rows).
  int qnode = t * kTileM + r;
  if (qnode >= M) qnode = M - 1;
  f32x4 qh[QL ? 1 : KS], ql[QL ? 1 : KS];
  float* qslot = nullptr;          // QL: this lane's 16-byte slots, [s8][plane] 1 KB apart
  f32x4 qraw[HK / 4];
  if constexpr (H16) {
    const _Float16* qp = as_h16(qkv) + ((size_t)qnode * B + b) * (size_t)D + head * DH + hh * HK;
#pragma unroll
    for (int s8 = 0; s8 < KS; ++s8) qraw[s8] = ld4(reinterpret_cast<const float*>(qp + 8 * s8));
  } else {
    const float* qp = qkv + ((size_t)qnode * B + b) * (3 * (size_t)D) + head * DH + hh * HK;
#pragma unroll
    for (int i = 0; i < HK / 4; ++i) qraw[i] = ld4(qp + 4 * i);
  }
  auto q_finish = [&]() __attribute__((always_inline)) {
    if constexpr (H16) {                         // the stored halfs are the fragments (the logits are scaled instead)
      if constexpr (QL) {
        extern __shared__ __attribute__((aligned(16))) int s_dyn_q[];
        qslot = reinterpret_cast<float*>(s_dyn_q) + max_chunks * 64 + (size_t)(blockDim.x >> 6) * (NP * 32 * DH / 2) +
                (size_t)head * (KS * NP * 256) + lane * 4;
#pragma unroll
        for (int s8 = 0; s8 < KS; ++s8) st4(qslot + (s8 * NP) * 256, qraw[s8]);
      } else {
#pragma unroll
        for (int s8 = 0; s8 < KS; ++s8) qh[s8] = qraw[s8];
      }
      return;
    }
    float qf[HK];
    const float qs = FEAT16 ? 1.0f : scale;   // FEAT16: q stays an exact fp16 value, the logits are scaled instead
#pragma unroll
    for (int i = 0; i < HK / 4; ++i) {
      qf[4 * i] = qraw[i][0] * qs; qf[4 * i + 1] = qraw[i][1] * qs; qf[4 * i + 2] = qraw[i][2] * qs; qf[4 * i + 3] = qraw[i][3] * qs;
    }
    if constexpr (QL) {
      extern __shared__ __attribute__((aligned(16))) int s_dyn_q[];
      qslot = reinterpret_cast<float*>(s_dyn_q) + max_chunks * 64 + (size_t)(blockDim.x >> 6) * (NP * 32 * DH / 2) +
              (size_t)head * (KS * NP * 256) + lane * 4;
#pragma unroll
      for (int s8 = 0; s8 < KS; ++s8) {
        f32x4 a, b2;
        if constexpr (FEAT16) split8_hi(qf + 8 * s8, a);
        else split8(qf + 8 * s8, a, b2);
        st4(qslot + (s8 * NP) * 256, a);
        if constexpr (!FEAT16) st4(qslot + (s8 * NP + 1) * 256, b2);
      }
    } else {
#pragma unroll
      for (int s8 = 0; s8 < KS; ++s8) {
        if constexpr (FEAT16) split8_hi(qf + 8 * s8, qh[s8]);
        else split8(qf + 8 * s8, qh[s8], ql[s8]);
      }
    }
  };
  f32x16 oacc[NS], oaccx[NS];
#pragma unroll
  for (int sl = 0; sl < NS; ++sl)
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      oacc[sl][g] = 0.f;
      oaccx[sl][g] = 0.f;
    }
  float m_run = kNegBig, l_run = 0.f;

  int lo, hi;
  if constexpr (ITEMS) {
    lo = it_lo;
    hi = it_hi;
  } else {
    const int c_begin = tile_chunk_start[t], nc = tile_chunk_start[t + 1] - c_begin;
    lo = c_begin + (nc * sp) / S;
    hi = c_begin + (nc * (sp + 1)) / S;
  }
  extern __shared__ __attribute__((aligned(16))) int s_dyn[];
  int* s_idx = s_dyn;                                               // [max_chunks * 32]
  unsigned* s_msk = reinterpret_cast<unsigned*>(s_dyn + max_chunks * 32);
  _Float16* vreg = reinterpret_cast<_Float16*>(s_dyn + max_chunks * 64) + (size_t)head * (NP * 32 * DH);   // this wave's V tile
  for (int i = threadIdx.x; i < (hi - lo) * 32; i += blockDim.x) {
    s_idx[i] = union_idx[lo * 32 + i];
    s_msk[i] = mask_bits[lo * 32 + i];
  }
  GC_ASTAMP(1);                                                      // q split + parked, index loads issued
  __syncthreads();
  GC_ASTAMP(2);                                                      // indices in LDS
  const size_t rstride = (size_t)B * 4 * D;                         // halfs between consecutive nodes in kv16
  const _Float16* kplane = kv16 + (size_t)b * 4 * D + head * DH + hh * HK;          // + node * rstride; lo plane at + D
  const _Float16* vplane = kv16 + (size_t)b * 4 * D + 2 * D + head * DH;            // + node * rstride + 8 * c8

  // V piece i of this lane: key vkey[i] of the chunk, 16-byte chunk vc8[i] of its row
  auto v_issue = [&](int c, f32x4 (&vr)[NP][NPV]) __attribute__((always_inline)) {
#if defined(GC_EXP_ATT_NOV)      // timing-only ablation: no V gathers at all
#pragma unroll
    for (int i = 0; i < NPV; ++i)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl)
#pragma unroll
        for (int j = 0; j < 4; ++j) vr[pl][i][j] = __int_as_float(0x3c003c00 + c + i + j);
    return;
#endif
#pragma unroll
    for (int i = 0; i < NPV; ++i) {
      const int p = lane + 64 * i, key = p / CPR, c8 = p - key * CPR;
      const _Float16* src = vplane + (size_t)(unsigned)s_idx[(c - lo) * 32 + key] * rstride + 8 * c8;
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) vr[pl][i] = ld4(reinterpret_cast<const float*>(src + pl * D));
    }
  };
  auto v_stage = [&](const f32x4 (&vr)[NP][NPV]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NPV; ++i) {
      const int p = lane + 64 * i, key = p / CPR, c8 = p - key * CPR;
#pragma unroll
      for (int pl = 0; pl < NP; ++pl)
        st4(reinterpret_cast<float*>(vreg + pl * (32 * DH) + key * DH + 8 * (c8 ^ v2_swz<DH>(key))), vr[pl][i]);
    }
  };
  auto k_issue = [&](int c, f32x4 (&kh)[KS], f32x4 (&kl)[KS]) __attribute__((always_inline)) {
    // GC_EXP_ATT_*: TIMING-ONLY ablations (wrong values) for tools/build_variant.sh + tools/exp_att_gather.sh; never defined
    // in the product build (profiles/r04_attention_gather_ablation.txt)
#if defined(GC_EXP_ATT_NOK)      // no K gathers at all: the operand registers are filled with constants
#pragma unroll
    for (int s8 = 0; s8 < KS; ++s8)
#pragma unroll
      for (int j = 0; j < 4; ++j) { kh[s8][j] = __int_as_float(0x3c003c00 + c + s8 + j); kl[s8][j] = __int_as_float(0x1c001c00 + c + s8 + j); }
    return;
#elif defined(GC_EXP_ATT_KCOAL)  // the same K bytes fetched quad-coalesced (4 lanes = 64 contiguous bytes)
    const _Float16* kb = kv16 + (size_t)b * 4 * D + head * DH;
#pragma unroll
    for (int s8 = 0; s8 < KS; ++s8) {
      const int key = (16 * (s8 & 1) + (lane >> 2)) & 31, piece = (4 * (s8 >> 1) + (lane & 3)) % (DH / 8);
      const _Float16* kq = kb + (size_t)(unsigned)s_idx[(c - lo) * 32 + key] * rstride + 8 * piece;
      kh[s8] = ld4(reinterpret_cast<const float*>(kq));
      if constexpr (!FEAT16) kl[s8] = ld4(reinterpret_cast<const float*>(kq + D));
    }
    return;
#endif
    const _Float16* kp = kplane + (size_t)(unsigned)s_idx[(c - lo) * 32 + r] * rstride;
#pragma unroll
    for (int s8 = 0; s8 < KS; ++s8) {
      kh[s8] = ld4(reinterpret_cast<const float*>(kp + 8 * s8));
      if constexpr (!FEAT16) kl[s8] = ld4(reinterpret_cast<const float*>(kp + D + 8 * s8));
    }
  };
  // transposed-read address of this lane: rows 4 hh + q of a 16-key group, 16 columns per 16-lane group
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
  const int trow = 4 * hh + tq;                                      // + 16 u + 8 t

  f32x4 kh[KS], kl[KS];
  if (lo < hi) {
    f32x4 v0[NP][NPV];
    v_issue(lo, v0);
    k_issue(lo, kh, kl);
    q_finish();                                  // behind the gathers just issued
    v_stage(v0);
  } else {
    q_finish();
  }
  GC_ASTAMP(3);                                                      // first chunk's V staged, K in flight
#ifdef GC_STAMPS
  tph = ast[3];
#endif
  for (int c = lo; c < hi; ++c) {
    // ---- next chunk's V goes out first; it lands behind this chunk's MFMAs ----
    f32x4 vn[NP][NPV];
    const int cn = (c + 1 < hi) ? c + 1 : c;
    v_issue(cn, vn);
    const unsigned mb = s_msk[(c - lo) * 32 + r];

    // ---- S^T = K . Q^T ----
    f32x16 st, stx;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      st[g] = 0.f;
      stx[g] = 0.f;
    }
#pragma unroll
    for (int s8 = 0; s8 < KS; ++s8) {
      f32x4 qa, qb;
      if constexpr (QL) {
        qa = ld4(qslot + (s8 * NP) * 256);
        if constexpr (!FEAT16) qb = ld4(qslot + (s8 * NP + 1) * 256);
      } else {
        qa = qh[s8];
        if constexpr (!FEAT16) qb = ql[s8];
      }
      if constexpr (!FEAT16) stx = mfma16(kh[s8], qb, stx);
      st = mfma16(kh[s8], qa, st);
      if constexpr (!FEAT16) stx = mfma16(kl[s8], qa, stx);
    }
    // the next chunk's K rows are fetched into the operand registers the MFMAs above have just read
    // (no second register set: the softmax and the P.V product below cover the loads)
    __builtin_amdgcn_sched_barrier(0);
    k_issue(cn, kh, kl);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      if constexpr (FEAT16) st[g] *= scale;
      else st[g] = hilo(st[g], stx[g]);
    }
    GC_ASTAMP_ACC(4, tph);                                           // sum over chunks: QK^T (incl. waiting for K)

    // ---- masked online softmax: the mask is applied ONCE (masked logits become -inf, whose exp is an exact
    // 0), so the maximum and the exponentials need no second select per element ----
    float cmax = kNegBig;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const bool on = (mb >> acc_row(g, hh)) & 1u;
      st[g] = on ? st[g] : -INFINITY;
      cmax = fmaxf(cmax, st[g]);
    }
    cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
    const bool need = cmax > m_run + kThr;
    if (__any(need)) {
      const float m_new = need ? cmax : m_run;
      const float alpha = __expf(m_run - m_new);
      l_run *= alpha;
      m_run = m_new;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const float af = __shfl(alpha, acc_row(g, hh));
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
          oacc[sl][g] *= af;
          oaccx[sl][g] *= af;
        }
      }
    }
    float pv[16];
    float psum = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      float p = __expf(st[g] - m_run);          // exp(-inf) = 0 for the masked keys
      if constexpr (FEAT16) p = r16(p);
      pv[g] = p;
      psum += p;
    }
    psum += __shfl_xor(psum, 32);
    l_run += psum;
    GC_ASTAMP_ACC(5, tph);                                           // softmax

    // ---- O += P . V : B operand = 8 keys of one dv column, by two transposed reads of the LDS tile ----
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f32x4 ph, pl;
      if constexpr (FEAT16) split8_hi(pv + 8 * u, ph);
      else split8(pv + 8 * u, ph, pl);
#pragma unroll
      for (int sl = 0; sl < NS; ++sl) {
        f32x4 vop[NP];
#pragma unroll
        for (int pn = 0; pn < NP; ++pn) {
          h4raw half[2];
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {
            const int row = trow + 16 * u + 8 * tt;
            const int chunk = (sl * 4 + tg * 2 + (tp >> 1)) ^ v2_swz<DH>(row);
            const _Float16* a = vreg + pn * (32 * DH) + row * DH + 8 * chunk + 4 * (tp & 1);
            half[tt] = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                (__attribute__((address_space(3))) h4raw*)(a));
          }
          vop[pn] = join_tr(half[0], half[1]);
        }
        if constexpr (FEAT16) {
          oacc[sl] = mfma16(ph, vop[0], oacc[sl]);
        } else {
          oaccx[sl] = mfma16(ph, vop[NP - 1], oaccx[sl]);
          oacc[sl] = mfma16(ph, vop[0], oacc[sl]);
          oaccx[sl] = mfma16(pl, vop[0], oaccx[sl]);
        }
      }
    }
    GC_ASTAMP_ACC(6, tph);                                           // P split + transposed reads + P.V issue
    // ---- the next chunk's V replaces this one in LDS (this wave's reads above were issued first) ----
    v_stage(vn);
    GC_ASTAMP_ACC(7, tph);                                           // next V staged (incl. waiting for its loads)
  }
  GC_ASTAMP(8);

  if (ITEMS ? (it_slot < 0) : (S == 1)) {
    const float inv_l = (l_run != 0.f) ? 1.0f / l_run : 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int qrow = acc_row(g, hh);
      const float il = __shfl(inv_l, qrow);
      const int node = t * kTileM + qrow;
      if (node < M) {
        const size_t orow = (size_t)node * B + b;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
          const float ov = (hilo(oacc[sl][g], oaccx[sl][g])) * il;
          if constexpr (H16) as_h16(o)[orow * D + head * DH + sl * 32 + r] = (_Float16)ov;
          else o[orow * D + head * DH + sl * 32 + r] = r16_if(ov, FEAT16 ? 1 : 0);
        }
      }
    }
  } else {
    const size_t slot = ((ITEMS ? (size_t)it_slot : (size_t)t * S + sp) * B + b) * H + head;
    float* po = part_o + slot * (kTileM * DH);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int qrow = acc_row(g, hh);
#pragma unroll
      for (int sl = 0; sl < NS; ++sl)
        po[qrow * DH + sl * 32 + r] = hilo(oacc[sl][g], oaccx[sl][g]);
    }
    if (hh == 0) {
      float* pm = part_ml + slot * (kTileM * 2);
      pm[r * 2] = m_run;
      pm[r * 2 + 1] = l_run;
    }
  }
#ifdef GC_STAMPS
  GC_ASTAMP(9);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  GC_ASTAMP(10);
  ast[11] = (unsigned long long)(hi - lo) + ((__builtin_amdgcn_s_memrealtime() - art0) << 16);   // chunks | 100-MHz ticks
  if (g_att_stamps && lane == 0) {
    unsigned long long* o = g_att_stamps + ((size_t)blockIdx.x * (blockDim.x >> 6) + head) * 12;
    for (int i = 0; i < 12; ++i) o[i] = ast[i];
  }
#endif
}

hipError_t launch_attention_v2(hipStream_t s, const float* qkv, const void* kv16, float* o, float* part_o,
                               float* part_ml, int M, int B, int D, int H, int S, const int* tile_chunk_start,
                               const int* union_idx, const unsigned* mask_bits, int n_tiles, int max_chunks,
                               bool feat16, bool h16, const int* items, int n_items) {
  if (H < 1 || D % H || S < 1 || !kv16 || max_chunks < 1 || (h16 && !feat16)) return hipErrorInvalidValue;
  const int dh = D / H;
  if ((dh != 32 && dh != 64 && dh != 128) || (dh == 128 && H > 4) || H > 8) return hipErrorInvalidValue;
  if (items && (S != 1 || n_items < 8 || n_items % 8 || dh != 128)) return hipErrorInvalidValue;   // item lists: the 1-degree form
  const dim3 grid(items ? n_items : ((n_tiles * S + 7) / 8) * 8, 1, B), block(64 * H);
  const int mc = (max_chunks + S - 1) / S + 1;
  const int np = feat16 ? 1 : 2;
  const size_t lds = (size_t)mc * 32 * 2 * sizeof(int) + (size_t)H * np * 32 * dh * sizeof(_Float16) +
                     (dh >= 64 ? (size_t)H * (dh / 16) * np * 1024 : 0);   // + the parked q fragments (heads >= 64)
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  const _Float16* kv = reinterpret_cast<const _Float16*>(kv16);
#define GC_ATT2I(DH_, F_, H_, I_, NT_)                                                                      \
  {                                                                                                         \
    static DynLdsOnce once;                                                                                 \
    if (hipError_t e = once.ensure((const void*)gc_attention_v2_kernel<DH_, F_, H_, I_>, 160 * 1024)) return e; \
    hipLaunchKernelGGL((gc_attention_v2_kernel<DH_, F_, H_, I_>), grid, block, lds, s, qkv, kv, o, part_o, part_ml, M, B, D, \
                       S, tile_chunk_start, union_idx, mask_bits, NT_, mc, items);                          \
  }
#define GC_ATT2(DH_, F_, H_) GC_ATT2I(DH_, F_, H_, false, n_tiles)
#define GC_ATT2F(DH_) { if (h16) GC_ATT2(DH_, true, true) else if (feat16) GC_ATT2(DH_, true, false) else GC_ATT2(DH_, false, false) }
  if (items) {
    if (h16) GC_ATT2I(128, true, true, true, n_items)
    else if (feat16) GC_ATT2I(128, true, false, true, n_items)
    else GC_ATT2I(128, false, false, true, n_items)
  } else if (dh == 32) GC_ATT2F(32)
  else if (dh == 64) GC_ATT2F(64)
  else GC_ATT2F(128)
#undef GC_ATT2F
#undef GC_ATT2
#undef GC_ATT2I
  return hipGetLastError();
}

// Merges the S partial (m, l, O) triples of every (node, head):
//   O = sum_s e^{m_s - m*} O_s / sum_s e^{m_s - m*} l_s,  m* = max_s m_s.
__global__ __launch_bounds__(256) void gc_attn_combine_kernel(const float* __restrict__ part_o,
                                                               const float* __restrict__ part_ml,
                                                               int M, int B, int D, int H, int S,
                                                               float* __restrict__ o, int out_s16, int round16) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);   // node * B + b
  const int lane = threadIdx.x & 63;
  if (row >= M * B) return;
  const int node = row / B, b = row - node * B;
  const int t = node / kTileM, q = node - t * kTileM;
  const int DH = D / H;
  for (int c = 4 * lane; c < D; c += 256) {
    const int head = c / DH, dv = c - head * DH;
    float mstar = -1e30f;
    for (int s = 0; s < S; ++s) {
      const size_t slot = (((size_t)t * S + s) * B + b) * H + head;
      mstar = fmaxf(mstar, part_ml[slot * (kTileM * 2) + q * 2]);
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float lsum = 0.f;
    for (int s = 0; s < S; ++s) {
      const size_t slot = (((size_t)t * S + s) * B + b) * H + head;
      const float m = part_ml[slot * (kTileM * 2) + q * 2];
      const float l = part_ml[slot * (kTileM * 2) + q * 2 + 1];
      const float w = (l != 0.f) ? expf(m - mstar) : 0.f;
      const float4 v = *reinterpret_cast<const float4*>(part_o + slot * (kTileM * DH) + q * DH + dv);
      acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
      lsum += w * l;
    }
    const float il = (lsum != 0.f) ? 1.0f / lsum : 0.f;
    float4 ov = make_float4(acc.x * il, acc.y * il, acc.z * il, acc.w * il);
    if (round16) { ov.x = r16(ov.x); ov.y = r16(ov.y); ov.z = r16(ov.z); ov.w = r16(ov.w); }
    if (out_s16) store4_s16(o, (size_t)row, D, c, ov.x, ov.y, ov.z, ov.w);
    else *reinterpret_cast<float4*>(o + (size_t)row * D + c) = ov;
  }
}

hipError_t launch_attention(hipStream_t s, const float* qkv, float* o, float* part_o, float* part_ml,
                            int M, int B, int D, int H, int S, bool out_s16, const int* tile_chunk_start,
                            const int* union_idx, const unsigned* mask_bits, int n_tiles, bool f16, int max_chunks,
                            bool feat16, const int* items, int n_items) {
  const int os = out_s16 ? 1 : 0;
  if (H < 1 || D % H || S < 1) return hipErrorInvalidValue;
  const int dh = D / H;
  if ((dh == 128 && H > 4) || H > 8) return hipErrorInvalidValue;
  if (items && (S != 1 || n_items < 8 || n_items % 8 || out_s16)) return hipErrorInvalidValue;
  if (!items) n_items = 0;
  dim3 grid(items ? n_items : n_tiles, S, B), block(64 * H);
  // (round 1's f16x3 form of this kernel, gc_attention16, re-split K and V per tile; it served only A/B switches since
  //  gc_attention_v2 and left the build in round 5: `f16` is accepted and ignored, this is the exact-f32 attention)
  (void)f16; (void)max_chunks;
  if (dh == 32)
    hipLaunchKernelGGL((gc_attention_kernel<32>), grid, block, 0, s, qkv, o, part_o, part_ml, M, B, D, S, os,
                       tile_chunk_start, union_idx, mask_bits, feat16 ? 1 : 0, items, n_items);
  else if (dh == 64)
    hipLaunchKernelGGL((gc_attention_kernel<64>), grid, block, 0, s, qkv, o, part_o, part_ml, M, B, D, S, os,
                       tile_chunk_start, union_idx, mask_bits, feat16 ? 1 : 0, items, n_items);
  else if (dh == 128)
    hipLaunchKernelGGL((gc_attention_kernel<128>), grid, block, 0, s, qkv, o, part_o, part_ml, M, B, D, S, os,
                       tile_chunk_start, union_idx, mask_bits, feat16 ? 1 : 0, items, n_items);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_attn_combine(hipStream_t s, const float* part_o, const float* part_ml, int M, int B,
                               int D, int H, int S, float* o, bool out_s16, bool round16) {
  hipLaunchKernelGGL(gc_attn_combine_kernel, dim3((M * B + 3) / 4), dim3(256), 0, s, part_o, part_ml, M,
                     B, D, H, S, o, out_s16 ? 1 : 0, round16 ? 1 : 0);
  return hipGetLastError();
}

// f16x3 domain guard: counts (at most once per workgroup) when `p` holds a non-finite value.
__global__ __launch_bounds__(256) void gc_finite_check_kernel(const float* __restrict__ p, size_t n,
                                                               unsigned* __restrict__ counter) {
  auto nonfinite = [](float v) { return (__builtin_bit_cast(unsigned, v) & 0x7f800000u) == 0x7f800000u; };
  unsigned bad = 0;
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 v = ld4(p + 4 * i);
#pragma unroll
    for (int e = 0; e < 4; ++e) bad |= nonfinite(v[e]);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) bad |= nonfinite(p[4 * n4 + threadIdx.x]);
  if (__any(bad != 0) && (threadIdx.x & 63) == 0) atomicAdd(counter, 1u);
}

hipError_t launch_finite_check(hipStream_t s, const float* p, size_t n, unsigned* counter) {
  const int grid = (int)std::min<size_t>((n / 4 + 255) / 256 + 1, 1024);
  hipLaunchKernelGGL(gc_finite_check_kernel, dim3(grid), dim3(256), 0, s, p, n, counter);
  return hipGetLastError();
}

// ----------------------------------------------------------------------------
// Grid-input packing and the sampler's elementwise updates
// (denoiser.py:654-659; dpm_solver_plus_plus_2s.py:139-154,181-205).
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gc_pack_full_kernel(const float* __restrict__ grid_struct,
                                                            const float* __restrict__ feats, int G,
                                                            int B, int c_in, int kp,
                                                            float* __restrict__ xp) {
  const size_t total = (size_t)G * B * kp;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t row = i / kp;
    const int c = (int)(i - row * kp);
    float v = 0.f;
    if (c < 3)
      v = grid_struct[(row / B) * 3 + c];
    else if (c < 3 + c_in)
      v = feats[row * c_in + (c - 3)];
    xp[i] = v;
  }
}

__global__ __launch_bounds__(256) void gc_write_noisy_kernel(const float* __restrict__ x,
                                                              const int* __restrict__ slots, int rows,
                                                              int c_out, int kp, float scale,
                                                              float* __restrict__ xp) {
  const size_t total = (size_t)rows * c_out;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t row = i / c_out;
    const int c = (int)(i - row * c_out);
    xp[row * kp + 3 + slots[c]] = scale * x[i];
  }
}

// The three kernels that produce the sampler's next state can also write it, scaled by the NEXT call's c_in, into the
// noisy-target slots of the packed grid input (what gc_write_noisy_kernel does as a launch of its own): nw.xp != nullptr.
__device__ __forceinline__ void put_noisy(const NoisyWrite& nw, size_t i, float v) {
  const size_t row = i / nw.c_out;
  const int c = (int)(i - row * nw.c_out);
  if (nw.xn) nw.xn[row * nw.ldn + c] = nw.scale * v;
  else nw.xp[row * nw.kp + 3 + nw.slots[c]] = nw.scale * v;
}

__global__ __launch_bounds__(256) void gc_write_noisy_compact_kernel(const float* __restrict__ x, int rows, int c_out,
                                                                      int ldn, float scale, float* __restrict__ xn) {
  const size_t total = (size_t)rows * c_out;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t row = i / c_out;
    xn[row * ldn + (i - row * c_out)] = scale * x[i];
  }
}

__global__ __launch_bounds__(256) void gc_zero_slots_kernel(const int* __restrict__ slots, int rows, int c_out, int kp,
                                                             float* __restrict__ xp) {
  const size_t total = (size_t)rows * c_out;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t row = i / c_out;
    xp[row * kp + 3 + slots[i - row * c_out]] = 0.f;
  }
}

__global__ __launch_bounds__(256) void gc_scale_kernel(const float* __restrict__ src, float a, size_t n,
                                                        float* __restrict__ dst, NoisyWrite nw) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float v = a * src[i];
    dst[i] = v;
    if (nw.xp || nw.xn) put_noisy(nw, i, v);
  }
}

__global__ __launch_bounds__(256) void gc_dpm_first_kernel(const float* __restrict__ y,
                                                            const float* __restrict__ x, float c_out,
                                                            float c_skip, float a_mid, size_t n,
                                                            float* __restrict__ den,
                                                            float* __restrict__ mid, NoisyWrite nw) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float xv = x[i];
    const float d = y[i] * c_out + xv * c_skip;
    den[i] = d;
    const float m = a_mid * xv + (1.0f - a_mid) * d;
    mid[i] = m;
    if (nw.xp || nw.xn) put_noisy(nw, i, m);
  }
}

__global__ __launch_bounds__(256) void gc_dpm_second_kernel(const float* __restrict__ y,
                                                             const float* __restrict__ xmid,
                                                             float c_out, float c_skip, float a_next,
                                                             size_t n, float* __restrict__ x, NoisyWrite nw) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float md = y[i] * c_out + xmid[i] * c_skip;
    const float v = a_next * x[i] + (1.0f - a_next) * md;
    x[i] = v;
    if (nw.xp || nw.xn) put_noisy(nw, i, v);
  }
}

__global__ __launch_bounds__(256) void gc_affine_rows_kernel(const float* __restrict__ src,
                                                              const float* __restrict__ cond,
                                                              int cond_stride, size_t items, int B,
                                                              int w, float* __restrict__ out, int round16, int h16) {
  const size_t total = items * B * w;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t row = i / w;
    const int c = (int)(i - row * w);
    const size_t item = row / B;
    const int b = (int)(row - item * B);
    const float* cs = cond + (size_t)b * cond_stride;
    const float v = src[item * w + c] * cs[c] + cs[w + c];
    if (h16) as_h16(out)[i] = (_Float16)v;      // physical fp16 storage
    else out[i] = r16_if(v, round16);
  }
}

static int ew_grid(size_t n) {
  size_t g = (n + 255) / 256;
  return (int)(g > 2048 ? 2048 : (g ? g : 1));
}

hipError_t launch_pack_full(hipStream_t s, const float* grid_struct, const float* feats, int G, int B,
                            int c_in, int kp, float* xp) {
  hipLaunchKernelGGL(gc_pack_full_kernel, dim3(ew_grid((size_t)G * B * kp)), dim3(256), 0, s,
                     grid_struct, feats, G, B, c_in, kp, xp);
  return hipGetLastError();
}

hipError_t launch_write_noisy(hipStream_t s, const float* x, const int* slots, int rows, int c_out,
                              int kp, float scale, float* xp) {
  hipLaunchKernelGGL(gc_write_noisy_kernel, dim3(ew_grid((size_t)rows * c_out)), dim3(256), 0, s, x,
                     slots, rows, c_out, kp, scale, xp);
  return hipGetLastError();
}

hipError_t launch_write_noisy_compact(hipStream_t s, const float* x, int rows, int c_out, int ldn, float scale, float* xn) {
  hipLaunchKernelGGL(gc_write_noisy_compact_kernel, dim3(ew_grid((size_t)rows * c_out)), dim3(256), 0, s, x, rows, c_out, ldn,
                     scale, xn);
  return hipGetLastError();
}

hipError_t launch_zero_slots(hipStream_t s, const int* slots, int rows, int c_out, int kp, float* xp) {
  hipLaunchKernelGGL(gc_zero_slots_kernel, dim3(ew_grid((size_t)rows * c_out)), dim3(256), 0, s, slots, rows, c_out, kp, xp);
  return hipGetLastError();
}

hipError_t launch_affine_rows(hipStream_t s, const float* src, const float* cond, int cond_stride,
                              int items, int B, int w, float* out, bool round16, bool h16) {
  hipLaunchKernelGGL(gc_affine_rows_kernel, dim3(ew_grid((size_t)items * B * w)), dim3(256), 0, s, src,
                     cond, cond_stride, (size_t)items, B, w, out, round16 ? 1 : 0, h16 ? 1 : 0);
  return hipGetLastError();
}

hipError_t launch_scale(hipStream_t s, const float* src, float a, size_t n, float* dst, const NoisyWrite& nw) {
  hipLaunchKernelGGL(gc_scale_kernel, dim3(ew_grid(n)), dim3(256), 0, s, src, a, n, dst, nw);
  return hipGetLastError();
}

hipError_t launch_dpm_first(hipStream_t s, const float* y, const float* x, float c_out, float c_skip,
                            float a_mid, size_t n, float* den, float* mid, const NoisyWrite& nw) {
  hipLaunchKernelGGL(gc_dpm_first_kernel, dim3(ew_grid(n)), dim3(256), 0, s, y, x, c_out, c_skip,
                     a_mid, n, den, mid, nw);
  return hipGetLastError();
}

hipError_t launch_dpm_second(hipStream_t s, const float* y, const float* xmid, float c_out,
                             float c_skip, float a_next, size_t n, float* x, const NoisyWrite& nw) {
  hipLaunchKernelGGL(gc_dpm_second_kernel, dim3(ew_grid(n)), dim3(256), 0, s, y, xmid, c_out, c_skip,
                     a_next, n, x, nw);
  return hipGetLastError();
}

// ----------------------------------------------------------------------------
// Autoregressive context update (training/train_helpers.py:596-622 +
// common/normalization.py:100-121 in normalised space): one thread per element of
// the new conditioning, one plan entry per channel.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gc_rollout_advance_kernel(
    const float* __restrict__ old_feats, const float* __restrict__ sample, const float* __restrict__ forcings,
    const int* __restrict__ kind, const int* __restrict__ src, const int* __restrict__ sidx,
    const float* __restrict__ a, const float* __restrict__ b, int rows, int c_in, int c_out, int n_forcing,
    float* __restrict__ new_feats) {
  const size_t total = (size_t)rows * c_in;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t row = i / c_in;
    const int c = (int)(i - row * c_in);
    const int k = kind[c];
    float v;
    if (k == 0) v = old_feats[i];
    else if (k == 1) v = old_feats[row * c_in + src[c]];
    else if (k == 2) v = old_feats[row * c_in + src[c]] + a[c] * sample[row * c_out + sidx[c]] + b[c];
    else if (k == 3) v = forcings[row * n_forcing + sidx[c]];
    else v = a[c] * sample[row * c_out + sidx[c]] + b[c];
    new_feats[i] = v;
  }
}

hipError_t launch_rollout_advance(hipStream_t s, const float* old_feats, const float* sample, const float* forcings,
                                  const int* kind, const int* src, const int* sidx, const float* a, const float* b,
                                  int rows, int c_in, int c_out, int n_forcing, float* new_feats) {
  const size_t total = (size_t)rows * c_in;
  const int grid = (int)std::min<size_t>((total + 255) / 256, 2048);
  hipLaunchKernelGGL(gc_rollout_advance_kernel, dim3(grid), dim3(256), 0, s, old_feats, sample, forcings, kind, src,
                     sidx, a, b, rows, c_in, c_out, n_forcing, new_feats);
  return hipGetLastError();
}

const char* kernel_class_name(int cls) {
  static const char* names[KC_COUNT] = {"gc_cond",      "gc_pack",       "gc_mlp",       "gc_segsum",
                                        "gc_rowop",     "gc_gemm_qkv",   "gc_attention", "gc_attn_combine",
                                        "gc_gemm_out",  "gc_gemm_ffw1",  "gc_gemm_ffw2", "gc_gemm_node",
                                        "gc_noise"};
  return (cls >= 0 && cls < KC_COUNT) ? names[cls] : "?";
}

}  // namespace GC_TU_NS
