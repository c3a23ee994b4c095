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

#include <math.h>

namespace gc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int acc_row(int g, int hh) { return (g & 3) + 8 * (g >> 2) + 4 * hh; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ float gelu_tanh(float x) {
  const float c = 0.7978845608028654f;  // sqrt(2/pi)
  return 0.5f * x * (1.0f + tanhf(c * (x + 0.044715f * x * x * x)));
}

__device__ __forceinline__ float swish(float x) { return x / (1.0f + expf(-x)); }

// acc[nt] += A[32 x K] * W[K x 32] for NT column tiles.
//   a_row : this lane's A row, already offset by hh*8  (LDS or global)
//   w_row : this lane's W^T row of column tile 0, already offset by hh*8 (global)
//   w_tile_stride : floats between consecutive 32-column tiles of W^T
template <int NT>
__device__ __forceinline__ void wave_gemm(f32x16 (&acc)[NT], const float* __restrict__ a_row,
                                          const float* __restrict__ w_row, size_t w_tile_stride,
                                          int K) {
  float4 bc[NT][2];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    bc[nt][0] = *reinterpret_cast<const float4*>(w_row + nt * w_tile_stride);
    bc[nt][1] = *reinterpret_cast<const float4*>(w_row + nt * w_tile_stride + 4);
  }
  for (int k0 = 0; k0 < K; k0 += 16) {
    const float4 a0 = *reinterpret_cast<const float4*>(a_row + k0);
    const float4 a1 = *reinterpret_cast<const float4*>(a_row + k0 + 4);
    const int kn = (k0 + 16 < K) ? k0 + 16 : k0;  // prefetch next chunk (re-reads the last one)
    float4 bn[NT][2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      bn[nt][0] = *reinterpret_cast<const float4*>(w_row + nt * w_tile_stride + kn);
      bn[nt][1] = *reinterpret_cast<const float4*>(w_row + nt * w_tile_stride + kn + 4);
    }
    const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float bv = (j < 4) ? ((const float*)&bc[nt][0])[j] : ((const float*)&bc[nt][1])[j - 4];
        acc[nt] = mfma32(av[j], bv, acc[nt]);
      }
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
constexpr int kChunkK = 256;         // columns of A staged in LDS per pass
constexpr int kLdA = kChunkK + 4;    // +4 floats: odd multiple of 16 B -> conflict-free ds_read_b128

template <int NT1, int NT2>
__global__ __launch_bounds__(256) void gc_mlp_kernel(MlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int hidden = NT1 * 128;
  const int ldh = hidden + 4;
  const int n_pad = NT2 * 128;
  const int ldy = n_pad + 4;
  const int a_floats = kTileM * ((kLdA > ldy) ? kLdA : ldy);
  float* bufA = smem;            // staged input chunk; later the pre-norm output tile
  float* bufH = smem + a_floats;  // hidden activations [32][hidden]

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int row0 = blockIdx.x * kTileM;

  f32x16 acc[NT1];
#pragma unroll
  for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[nt][g] = 0.f;

  int koff = 0;
  for (int s = 0; s < a.nseg; ++s) {
    const Segment sg = a.seg[s];
    for (int c0 = 0; c0 < sg.width; c0 += kChunkK) {
      const int kc = (sg.width - c0 < kChunkK) ? sg.width - c0 : kChunkK;
      const int n4 = kc >> 2;
      __syncthreads();  // the previous chunk has been consumed by every wave
      for (int idx = tid; idx < kTileM * n4; idx += 256) {
        const int row = idx / n4, c4 = idx - row * n4;
        int grow = row0 + row;
        if (grow >= a.rows) grow = a.rows - 1;
        const int item = grow / a.B, b = grow - item * a.B;
        size_t srow = sg.index ? (size_t)sg.index[item] : (size_t)item;
        if (!sg.bcast) srow = srow * a.B + b;
        float4 v = *reinterpret_cast<const float4*>(sg.ptr + srow * sg.ld + c0 + 4 * c4);
        if (sg.affine) {
          const float* sc = sg.affine + (size_t)b * a.cond_stride + c0 + 4 * c4;
          const float4 s4 = *reinterpret_cast<const float4*>(sc);
          const float4 o4 = *reinterpret_cast<const float4*>(sc + sg.width);
          v.x = v.x * s4.x + o4.x;
          v.y = v.y * s4.y + o4.y;
          v.z = v.z * s4.z + o4.z;
          v.w = v.w * s4.w + o4.w;
        }
        *reinterpret_cast<float4*>(bufA + row * kLdA + 4 * c4) = v;
      }
      __syncthreads();
      const float* w_row = a.w1t + (size_t)(wave * NT1 * 32 + r) * a.ldw1 + koff + c0 + hh * 8;
      wave_gemm<NT1>(acc, bufA + r * kLdA + hh * 8, w_row, (size_t)32 * a.ldw1, kc);
    }
    koff += sg.width;
  }

  // hidden = swish(acc + b1) -> LDS
#pragma unroll
  for (int nt = 0; nt < NT1; ++nt) {
    const int col = wave * NT1 * 32 + nt * 32 + r;
    const float bias = a.b1[col];
#pragma unroll
    for (int g = 0; g < 16; ++g) bufH[acc_row(g, hh) * ldh + col] = swish(acc[nt][g] + bias);
  }
  __syncthreads();

  f32x16 acc2[NT2];
#pragma unroll
  for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
    for (int g = 0; g < 16; ++g) acc2[nt][g] = 0.f;
  {
    const float* w_row = a.w2t + (size_t)(wave * NT2 * 32 + r) * hidden + hh * 8;
    wave_gemm<NT2>(acc2, bufH + r * ldh + hh * 8, w_row, (size_t)32 * hidden, hidden);
  }
#pragma unroll
  for (int nt = 0; nt < NT2; ++nt) {
    const int col = wave * NT2 * 32 + nt * 32 + r;
    const float bias = a.b2[col];
#pragma unroll
    for (int g = 0; g < 16; ++g) bufA[acc_row(g, hh) * ldy + col] = acc2[nt][g] + bias;
  }
  __syncthreads();

  // epilogue: each wave finishes 8 rows; lanes stride the columns.
  const int n = a.n_out;
  const float inv_n = 1.0f / (float)n;
  for (int rr = 0; rr < kTileM / 4; ++rr) {
    const int row = wave * (kTileM / 4) + rr;
    const int grow = row0 + row;
    if (grow >= a.rows) break;
    const float* y = bufA + row * ldy;
    float mean = 0.f, rstd = 1.f;
    if (a.do_ln) {
      float s1 = 0.f, s2 = 0.f;
      for (int c = lane; c < n; c += 64) {
        const float v = y[c];
        s1 += v;
        s2 += v * v;
      }
      s1 = wave_sum(s1);
      s2 = wave_sum(s2);
      mean = s1 * inv_n;
      const float var = fmaxf(s2 * inv_n - mean * mean, 0.f);
      rstd = 1.0f / sqrtf(var + 1e-6f);
    }
    const int b = grow % a.B;
    const float* cs = a.cond ? a.cond + (size_t)b * a.cond_stride : nullptr;
    for (int c = lane; c < n; c += 64) {
      float v = (y[c] - mean) * rstd;
      if (cs) v = v * cs[c] + cs[n + c];
      if (a.residual) v += a.residual[(size_t)grow * n + c];
      a.out[(size_t)grow * a.ldo + c] = v;
    }
  }
}

template <int NT1, int NT2>
static hipError_t launch_mlp_t(hipStream_t s, const MlpArgs& a) {
  const int hidden = NT1 * 128, n_pad = NT2 * 128;
  const int ldy = n_pad + 4;
  const int a_floats = kTileM * ((kLdA > ldy) ? kLdA : ldy);
  const size_t lds = (size_t)(a_floats + kTileM * (hidden + 4)) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gc_mlp_kernel<NT1, NT2>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int grid = (a.rows + kTileM - 1) / kTileM;
  hipLaunchKernelGGL((gc_mlp_kernel<NT1, NT2>), dim3(grid), dim3(256), lds, s, a);
  return hipGetLastError();
}

hipError_t launch_mlp(hipStream_t s, const MlpArgs& a) {
  const int nt1 = a.hidden / 128, nt2 = a.n_out_pad / 128;
  if (a.hidden % 128 || a.n_out_pad % 128) return hipErrorInvalidValue;
  if (nt1 == 1 && nt2 == 1) return launch_mlp_t<1, 1>(s, a);
  if (nt1 == 2 && nt2 == 2) return launch_mlp_t<2, 2>(s, a);
  if (nt1 == 2 && nt2 == 1) return launch_mlp_t<2, 1>(s, a);
  if (nt1 == 4 && nt2 == 4) return launch_mlp_t<4, 4>(s, a);
  if (nt1 == 4 && nt2 == 1) return launch_mlp_t<4, 1>(s, a);
  return hipErrorInvalidValue;
}

// ----------------------------------------------------------------------------
// gc_segsum: jraph.segment_sum by receiver (deep_typed_graph_net.py:396-410;
// typed_graph_net.py:175-182) as a CSR walk: one wave per output row, edges
// added in ascending edge id -> deterministic, no atomics.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gc_segsum_kernel(const float* __restrict__ src,
                                                         const int* __restrict__ rowptr,
                                                         const int* __restrict__ eids, int n_items,
                                                         int B, int width, float* __restrict__ out) {
  const int wrow = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (wrow >= n_items * B) return;
  const int item = wrow / B, b = wrow - item * B;
  const int e0 = rowptr[item], e1 = rowptr[item + 1];
  for (int c = lane * 4; c < width; c += 256) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int e = e0; e < e1; ++e) {
      const float4 v =
          *reinterpret_cast<const float4*>(src + ((size_t)eids[e] * B + b) * width + c);
      s.x += v.x;
      s.y += v.y;
      s.z += v.z;
      s.w += v.w;
    }
    *reinterpret_cast<float4*>(out + (size_t)wrow * width + c) = s;
  }
}

hipError_t launch_segsum(hipStream_t s, const float* src, const int* rowptr, const int* eids,
                         int n_items, int B, int width, float* out) {
  const int grid = (n_items * B + 3) / 4;
  hipLaunchKernelGGL(gc_segsum_kernel, dim3(grid), dim3(256), 0, s, src, rowptr, eids, n_items, B,
                     width, out);
  return hipGetLastError();
}

// ----------------------------------------------------------------------------
// gc_ln_gemm: out = act(cond(LayerNorm(x)) @ W + b) for a 32-row tile and a
// 4*NT*32-column slice.  Transformer pre-norm + projection
// (sparse_transformer.py:518-519 + 271-290 for QKV; :522-523 + 252-268 for FFW-1).
// The normalised, conditioned rows live only in LDS.
// ----------------------------------------------------------------------------
template <int NT, int CLS>
__global__ __launch_bounds__(256) void gc_ln_gemm_kernel(const float* __restrict__ x, int rows, int d,
                                                          int B, const float* __restrict__ cond,
                                                          int cond_stride,
                                                          const float* __restrict__ wt,
                                                          const float* __restrict__ bias, int n,
                                                          int act, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lda = d + 4;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int row0 = blockIdx.x * kTileM;
  const int n0 = blockIdx.y * (4 * NT * 32);
  const float inv_d = 1.0f / (float)d;

  for (int rr = 0; rr < kTileM / 4; ++rr) {
    const int row = wave * (kTileM / 4) + rr;
    int grow = row0 + row;
    if (grow >= rows) grow = rows - 1;
    const float* xr = x + (size_t)grow * d;
    float v[8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = lane + 64 * i;
      v[i] = (c < d) ? xr[c] : 0.f;
      s1 += v[i];
      s2 += v[i] * v[i];
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    const float mean = s1 * inv_d;
    const float var = fmaxf(s2 * inv_d - mean * mean, 0.f);
    const float rstd = 1.0f / sqrtf(var + 1e-6f);
    const float* cs = cond + (size_t)(grow % B) * cond_stride;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = lane + 64 * i;
      if (c < d) smem[row * lda + c] = (v[i] - mean) * rstd * cs[c] + cs[d + c];
    }
  }
  __syncthreads();

  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[nt][g] = 0.f;
  const int col0 = n0 + wave * NT * 32;
  wave_gemm<NT>(acc, smem + r * lda + hh * 8, wt + (size_t)(col0 + r) * d + hh * 8, (size_t)32 * d, d);

#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = col0 + nt * 32 + r;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int grow = row0 + acc_row(g, hh);
      if (grow < rows) {
        float v = acc[nt][g] + bv;
        if (act) v = gelu_tanh(v);
        out[(size_t)grow * n + col] = v;
      }
    }
  }
}

template <int CLS>
static hipError_t launch_ln_gemm_c(hipStream_t s, const float* x, int rows, int d, int B,
                                   const float* cond, int cond_stride, const float* wt,
                                   const float* bias, int n, int act, float* out) {
  if (d > 512 || d % 16 || n % 128) return hipErrorInvalidValue;
  const size_t lds = (size_t)kTileM * (d + 4) * sizeof(float);
  const int mt = (rows + kTileM - 1) / kTileM;
  if (n % 256 == 0) {
    static bool attr = false;
    if (!attr) {
      hipError_t e = hipFuncSetAttribute((const void*)gc_ln_gemm_kernel<2, CLS>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)(32 * 516 * 4));
      if (e != hipSuccess) return e;
      attr = true;
    }
    hipLaunchKernelGGL((gc_ln_gemm_kernel<2, CLS>), dim3(mt, n / 256), dim3(256), lds, s, x, rows, d, B,
                       cond, cond_stride, wt, bias, n, act, out);
  } else {
    static bool attr = false;
    if (!attr) {
      hipError_t e = hipFuncSetAttribute((const void*)gc_ln_gemm_kernel<1, CLS>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)(32 * 516 * 4));
      if (e != hipSuccess) return e;
      attr = true;
    }
    hipLaunchKernelGGL((gc_ln_gemm_kernel<1, CLS>), dim3(mt, n / 128), dim3(256), lds, s, x, rows, d, B,
                       cond, cond_stride, wt, bias, n, act, out);
  }
  return hipGetLastError();
}

hipError_t launch_ln_gemm(hipStream_t s, int cls, const float* x, int rows, int d, int B,
                          const float* cond, int cond_stride, const float* wt, const float* bias,
                          int n, int act, float* out) {
  if (cls == KC_LN_GEMM_QKV)
    return launch_ln_gemm_c<KC_LN_GEMM_QKV>(s, x, rows, d, B, cond, cond_stride, wt, bias, n, act, out);
  return launch_ln_gemm_c<KC_LN_GEMM_FFW1>(s, x, rows, d, B, cond, cond_stride, wt, bias, n, act, out);
}

// ----------------------------------------------------------------------------
// gc_gemm_res: out = res + a @ W + b on a 32 x (NT*32) tile; the 4 waves split K
// and their partial tiles are summed through LDS in a fixed order.
// Attention output projection + residual (sparse_transformer.py:351-353,:520) and
// FFW layer 2 + residual (:252-268,:524).
// ----------------------------------------------------------------------------
template <int NT, int CLS>
__global__ __launch_bounds__(256) void gc_gemm_res_kernel(const float* __restrict__ a, int rows, int k,
                                                           const float* __restrict__ wt,
                                                           const float* __restrict__ bias, int n,
                                                           const float* __restrict__ res,
                                                           float* __restrict__ out) {
  constexpr int TN = NT * 32;
  constexpr int LDR = TN + 1;
  __shared__ float red[4][kTileM][LDR];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int row0 = blockIdx.x * kTileM;
  const int n0 = blockIdx.y * TN;
  const int kq = k >> 2;  // K range of this wave
  int arow = row0 + r;
  if (arow >= rows) arow = rows - 1;

  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[nt][g] = 0.f;
  wave_gemm<NT>(acc, a + (size_t)arow * k + wave * kq + hh * 8,
                wt + (size_t)(n0 + r) * k + wave * kq + hh * 8, (size_t)32 * k, kq);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int g = 0; g < 16; ++g) red[wave][acc_row(g, hh)][nt * 32 + r] = acc[nt][g];
  __syncthreads();
  for (int idx = tid; idx < kTileM * TN; idx += 256) {
    const int row = idx / TN, c = idx - row * TN;
    const int grow = row0 + row;
    if (grow < rows) {
      const float sum = ((red[0][row][c] + red[1][row][c]) + red[2][row][c]) + red[3][row][c];
      const size_t o = (size_t)grow * n + n0 + c;
      out[o] = res[o] + (sum + bias[n0 + c]);
    }
  }
}

template <int CLS>
static hipError_t launch_gemm_res_c(hipStream_t s, const float* a, int rows, int k, const float* wt,
                                    const float* bias, int n, const float* res, float* out) {
  if (k % 64 || n % 32) return hipErrorInvalidValue;
  const int mt = (rows + kTileM - 1) / kTileM;
  hipLaunchKernelGGL((gc_gemm_res_kernel<1, CLS>), dim3(mt, n / 32), dim3(256), 0, s, a, rows, k, wt,
                     bias, n, res, out);
  return hipGetLastError();
}

hipError_t launch_gemm_res(hipStream_t s, int cls, const float* a, int rows, int k, const float* wt,
                           const float* bias, int n, const float* res, float* out) {
  if (cls == KC_GEMM_RES_OUT)
    return launch_gemm_res_c<KC_GEMM_RES_OUT>(s, a, rows, k, wt, bias, n, res, out);
  return launch_gemm_res_c<KC_GEMM_RES_FFW2>(s, a, rows, k, wt, bias, n, res, out);
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
__global__ __launch_bounds__(512) void gc_attention_kernel(
    const float* __restrict__ qkv, float* __restrict__ o, int M, int B, int D,
    const int* __restrict__ tile_chunk_start, const int* __restrict__ union_idx,
    const unsigned* __restrict__ mask_bits) {
  constexpr int HK = DH / 2;   // k-steps of the QK^T product (2 per MFMA across lane halves)
  constexpr int NS = DH / 32;  // 32-wide dv slices
  const int t = blockIdx.x, b = blockIdx.y;
  const int head = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
  const size_t ld = (size_t)3 * D;
  const float scale = 1.0f / sqrtf((float)DH);
  const float kNegBig = -1e30f;
  const float kThr = 40.0f;  // lazy-rescale threshold: p <= e^40, far inside f32 range

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

  const int c_begin = tile_chunk_start[t], c_end = tile_chunk_start[t + 1];
  for (int c = c_begin; c < c_end; ++c) {
    // ---- S^T = K . Q^T ------------------------------------------------------
    const int kidx = union_idx[c * 32 + r];
    const float* kp = qkv + ((size_t)kidx * B + b) * ld + D + head * DH + hh * HK;
    float kf[HK];
#pragma unroll
    for (int i = 0; i < HK; i += 4) {
      const float4 v = *reinterpret_cast<const float4*>(kp + i);
      kf[i] = v.x;
      kf[i + 1] = v.y;
      kf[i + 2] = v.z;
      kf[i + 3] = v.w;
    }
    // V row indices this lane-half will need: key(g, hh) = (g&3) + 8*(g>>2) + 4*hh
    int vidx[16];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int4 v = *reinterpret_cast<const int4*>(union_idx + c * 32 + 8 * q4 + 4 * hh);
      vidx[4 * q4] = v.x;
      vidx[4 * q4 + 1] = v.y;
      vidx[4 * q4 + 2] = v.z;
      vidx[4 * q4 + 3] = v.w;
    }
    f32x16 st;
#pragma unroll
    for (int g = 0; g < 16; ++g) st[g] = 0.f;
#pragma unroll
    for (int j = 0; j < HK; ++j) st = mfma32(kf[j], qf[j], st);

    // ---- masked online softmax (query = this lane's r; keys in registers) ----
    const unsigned mb = mask_bits[c * 32 + r];
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
      const float alpha = expf(m_run - m_new);  // 1 where nothing changed, 0 on first use
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
      const float p = on ? expf(st[g] - m_run) : 0.f;
      st[g] = p;
      psum += p;
    }
    psum += __shfl_xor(psum, 32);
    l_run += psum;

    // ---- O += P . V ----------------------------------------------------------
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const float* vp = qkv + ((size_t)vidx[g] * B + b) * ld + 2 * D + head * DH + r;
#pragma unroll
      for (int sl = 0; sl < NS; ++sl) oacc[sl] = mfma32(st[g], vp[sl * 32], oacc[sl]);
    }
  }

  const float inv_l = (l_run > 0.f) ? 1.0f / l_run : 0.f;
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int qrow = acc_row(g, hh);
    const float il = __shfl(inv_l, qrow);
    const int node = t * kTileM + qrow;
    if (node < M) {
      float* op = o + ((size_t)node * B + b) * D + head * DH + r;
#pragma unroll
      for (int sl = 0; sl < NS; ++sl) op[sl * 32] = oacc[sl][g] * il;
    }
  }
}

hipError_t launch_attention(hipStream_t s, const float* qkv, float* o, int M, int B, int D, int H,
                            const int* tile_chunk_start, const int* union_idx,
                            const unsigned* mask_bits, int n_tiles) {
  if (H < 1 || H > 8 || D % H) return hipErrorInvalidValue;
  const int dh = D / H;
  dim3 grid(n_tiles, B), block(64 * H);
  if (dh == 32)
    hipLaunchKernelGGL((gc_attention_kernel<32>), grid, block, 0, s, qkv, o, M, B, D, tile_chunk_start,
                       union_idx, mask_bits);
  else if (dh == 64)
    hipLaunchKernelGGL((gc_attention_kernel<64>), grid, block, 0, s, qkv, o, M, B, D, tile_chunk_start,
                       union_idx, mask_bits);
  else if (dh == 128)
    hipLaunchKernelGGL((gc_attention_kernel<128>), grid, block, 0, s, qkv, o, M, B, D,
                       tile_chunk_start, union_idx, mask_bits);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

// ----------------------------------------------------------------------------
// gc_ln_cond: final LayerNorm + conditioning of the transformer
// (sparse_transformer.py:630-633).  One wave per row.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gc_ln_cond_kernel(const float* __restrict__ x, int rows, int d,
                                                          int B, const float* __restrict__ cond,
                                                          int cond_stride, float* __restrict__ out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (size_t)row * d;
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane; c < d; c += 64) {
    const float v = xr[c];
    s1 += v;
    s2 += v * v;
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  const float mean = s1 / (float)d;
  const float var = fmaxf(s2 / (float)d - mean * mean, 0.f);
  const float rstd = 1.0f / sqrtf(var + 1e-6f);
  const float* cs = cond + (size_t)(row % B) * cond_stride;
  for (int c = lane; c < d; c += 64) out[(size_t)row * d + c] = (xr[c] - mean) * rstd * cs[c] + cs[d + c];
}

hipError_t launch_ln_cond(hipStream_t s, const float* x, int rows, int d, int B, const float* cond,
                          int cond_stride, float* out) {
  hipLaunchKernelGGL(gc_ln_cond_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, rows, d, B, cond,
                     cond_stride, out);
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

__global__ __launch_bounds__(256) void gc_scale_kernel(const float* __restrict__ src, float a, size_t n,
                                                        float* __restrict__ dst) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    dst[i] = a * src[i];
}

__global__ __launch_bounds__(256) void gc_dpm_first_kernel(const float* __restrict__ y,
                                                            const float* __restrict__ x, float c_out,
                                                            float c_skip, float a_mid, size_t n,
                                                            float* __restrict__ den,
                                                            float* __restrict__ mid) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float xv = x[i];
    const float d = y[i] * c_out + xv * c_skip;
    den[i] = d;
    mid[i] = a_mid * xv + (1.0f - a_mid) * d;
  }
}

__global__ __launch_bounds__(256) void gc_dpm_second_kernel(const float* __restrict__ y,
                                                             const float* __restrict__ xmid,
                                                             float c_out, float c_skip, float a_next,
                                                             size_t n, float* __restrict__ x) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float md = y[i] * c_out + xmid[i] * c_skip;
    x[i] = a_next * x[i] + (1.0f - a_next) * md;
  }
}

__global__ __launch_bounds__(256) void gc_affine_rows_kernel(const float* __restrict__ src,
                                                              const float* __restrict__ cond,
                                                              int cond_stride, size_t items, int B,
                                                              int w, float* __restrict__ out) {
  const size_t total = items * B * w;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t row = i / w;
    const int c = (int)(i - row * w);
    const size_t item = row / B;
    const int b = (int)(row - item * B);
    const float* cs = cond + (size_t)b * cond_stride;
    out[i] = src[item * w + c] * cs[c] + cs[w + c];
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

hipError_t launch_affine_rows(hipStream_t s, const float* src, const float* cond, int cond_stride,
                              int items, int B, int w, float* out) {
  hipLaunchKernelGGL(gc_affine_rows_kernel, dim3(ew_grid((size_t)items * B * w)), dim3(256), 0, s, src,
                     cond, cond_stride, (size_t)items, B, w, out);
  return hipGetLastError();
}

hipError_t launch_scale(hipStream_t s, const float* src, float a, size_t n, float* dst) {
  hipLaunchKernelGGL(gc_scale_kernel, dim3(ew_grid(n)), dim3(256), 0, s, src, a, n, dst);
  return hipGetLastError();
}

hipError_t launch_dpm_first(hipStream_t s, const float* y, const float* x, float c_out, float c_skip,
                            float a_mid, size_t n, float* den, float* mid) {
  hipLaunchKernelGGL(gc_dpm_first_kernel, dim3(ew_grid(n)), dim3(256), 0, s, y, x, c_out, c_skip,
                     a_mid, n, den, mid);
  return hipGetLastError();
}

hipError_t launch_dpm_second(hipStream_t s, const float* y, const float* xmid, float c_out,
                             float c_skip, float a_next, size_t n, float* x) {
  hipLaunchKernelGGL(gc_dpm_second_kernel, dim3(ew_grid(n)), dim3(256), 0, s, y, xmid, c_out, c_skip,
                     a_next, n, x);
  return hipGetLastError();
}

const char* kernel_class_name(int cls) {
  static const char* names[KC_COUNT] = {"gc_cond",        "gc_pack",         "gc_mlp",
                                        "gc_segsum",      "gc_ln_gemm_qkv",  "gc_attention",
                                        "gc_gemm_res_out", "gc_ln_gemm_ffw1", "gc_gemm_res_ffw2",
                                        "gc_ln_cond"};
  return (cls >= 0 && cls < KC_COUNT) ? names[cls] : "?";
}

}  // namespace gc
