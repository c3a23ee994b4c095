// Host-callable launchers of the gfx950 kernels (implemented in gc_kernels.hip).
// All pointers are device pointers; every launch goes to `stream` and returns
// immediately.  Row index convention: row = item * B + b  (item = node or edge).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gc {

// Kernel classes, used for per-class profiling and as kernel-name prefixes.
enum KernelClass : int {
  KC_COND = 0,      // noise-level encoder + all conditioning vectors
  KC_PACK,          // grid input packing / sampler elementwise
  KC_MLP,           // fused GNN MLP (+LN +cond +residual)
  KC_SEGSUM,        // CSR segment sum
  KC_ROWOP,         // residual add (+ split-K slab sum) + LayerNorm + cond of the mesh rows
  KC_GEMM_QKV,      // QKV projection
  KC_ATTN,          // k-hop sparse attention
  KC_ATTN_COMBINE,  // merge of the attention key-splits
  KC_GEMM_OUT,      // attention out-projection (slab)
  KC_GEMM_FFW1,     // FFW layer 1 + gelu
  KC_GEMM_FFW2,     // FFW layer 2 (split-K slabs)
  KC_GEMM_NODE,     // per-node halves of the edge MLPs' first layer
  KC_NOISE,         // spherical white noise: Philox normals + Legendre / Fourier synthesis
  KC_COUNT
};

constexpr int kTileM = 32;          // rows per block tile (one 32x32 MFMA tile high)
constexpr int kCondDim = 16;

// One source of input columns for the fused MLP (concatenated along K).
struct Segment {
  const float* ptr;    // source rows
  const int* index;    // per-item gather index, or nullptr (identity)
  const float* affine; // per-batch [scale(width) | offset(width)] applied on load, or nullptr
  int width;           // columns, multiple of 32
  int ld;              // source row stride (floats)
  int bcast;           // 1: source has no batch axis (row = item), 0: row = item*B + b
};

// A per-node product gathered per row and added to the first layer's pre-activation.
struct AddTerm {
  const float* ptr;    // [nodes * B][hidden] float32
  const int* index;    // per-item node index
};

struct MlpArgs {
  Segment seg[3];
  int nseg;
  AddTerm add[2];
  int nadd;
  int rows;            // items * B
  int B;
  int hidden;          // 128 / 256 / 512
  const float* w1t;    // [hidden][ldw1]  (transposed, K padded)
  int ldw1;
  const float* b1;     // [hidden]
  const float* w2t;    // [n_out_pad][hidden]
  const float* b2;     // [n_out_pad]
  int n_out;           // real output columns
  int n_out_pad;       // multiple of 128
  int do_ln;
  const float* cond;   // per-batch [scale(n_out) | offset(n_out)], or nullptr
  int cond_stride;     // floats between batch elements in cond / affine buffers
  const float* residual; // [rows][n_out] or nullptr
  float* out;          // [rows][ldo]
  int ldo;
  int f16;             // 1: w1t / w2t are S16-encoded; inputs are split on the fly (f16x3 products)
  // weight-streaming form (f16x3 only; used when w1f != nullptr and there are no add terms):
  const float* w1f;    // WF16 fragment-order image of W1^T, K zero-padded to k1f
  int k1f;             // padded K of w1f, multiple of 64, >= sum of segment widths
  const float* w2f;    // WF16 image of W2^T [n_out_pad][hidden]
  const float* ones;   // >= 512 ones / zeros: the identity affine of segments without one
  const float* zeros;
  // "fp16 node features" mode (gc_set_option features=f16; rounding points: DESIGN.md section 3b):
  int round16;         // 1: staged inputs, the hidden activation and the second Linear's output are rounded to fp16
  int round_out;       // 1: the LayerNorm + conditioning output and the residual sum are rounded to fp16
  int wt;              // 1: the output rows leave as write-through (sc1) stores (A/B switch GC_TUNE_WT_STORES & 2)
};

hipError_t launch_cond(hipStream_t s, const float* sigma_dev, float sigma_scalar, int B,
                       const float* w0t, const float* b0, const float* w1t, const float* b1,
                       int nfreq, int nhid, float base_period,
                       const float* wc_all, const float* bc_all, int total, float* cond_vec,
                       float* cond_out);

// Conditioning of up to kMaxSigmaList denoiser calls in one launch (sampler): cond_out[call][b][total]
constexpr int kMaxSigmaList = 96;
struct SigmaList { float v[kMaxSigmaList]; };
hipError_t launch_cond_multi(hipStream_t s, const SigmaList& sl, int ncalls, int B, const float* w0t,
                             const float* b0, const float* w1t, const float* b1, int nfreq, int nhid,
                             float base_period, const float* wc_all, const float* bc_all, int total,
                             float* cond_out);

hipError_t launch_mlp(hipStream_t s, const MlpArgs& a);

hipError_t launch_segsum(hipStream_t s, const float* src, const int* rowptr, const int* eids,
                         int n_items, int n_edges, int B, int width, float* out, bool round16 = false);

struct GemmArgs {
  const float* a;      // [rows][lda]
  int lda;
  int a_f32;           // f16 mode only: 1 = A holds plain float32 and is split to S16 while staging
  // att_S > 0: A is the attention output merged on the fly from att_S key-split partials
  // (launch_attention's part_o / part_ml); `a` is then unused.  Only with shape 1, epi 1.
  const float* att_po;
  const float* att_pml;
  int att_S, att_B, att_H, att_DH;
  const float* wt;     // W^T: [n][ldw]
  int ldw;
  int rows, n;
  int k_slice;         // K handled by one split (multiple of 32); split z covers [z*k_slice, (z+1)*k_slice)
  int splits;          // filled in by launch_gemm
  const float* bias;   // [n] or nullptr (epi 0)
  int act;             // 1: gelu(tanh) (epi 0)
  float* out;          // epi 0: [rows][ldo]; epi 1: slabs [splits][rows][ldo]
  int ldo;
  int round16;         // fp16-feature mode: epi 0 outputs and the merged attention output are rounded to fp16
  // epi 3 (QKV projection, weight-streaming form): besides out [rows][3d] f32, K and V are written
  // ALREADY SPLIT as fp16 planes for the attention kernel: kv16[row] = [K_hi(d) | K_lo(d) | V_hi(d) | V_lo(d)]
  void* kv16;
  int kv_d;            // d_model (n == 3 * kv_d)
};
// shape: 1 -> 32x128 tiles, 2 -> 64x128 tiles (256 threads);
// epi 0: bias/act f32 store, 1: raw split-K slabs (f32),
// 2: bias/act store in S16 split-fp16 layout (f16 only).  f16: A and W^T are S16-encoded and the
// product runs as 3 fp16 MFMAs per k-step (f32-equivalent accuracy, see gc_kernels.hip).
hipError_t launch_gemm(hipStream_t s, int cls, const GemmArgs& g, int shape, int splits, int epi, bool f16);
// Weight-streaming f16x3 form: g.wt is the WF16 fragment-order image of W^T (gc_api.hip
// encode_wf16), g.ldw the full contraction length K; a is plain float32 (or attention partials).
// Tile (32*mt) x 128; needs n % 128 == 0 and k_slice a multiple of 128.  epi 0 | 1 as launch_gemm;
// epi 3 = QKV projection with pre-split K / V planes (g.kv16, g.kv_d).
hipError_t launch_gemm_ws(hipStream_t s, int cls, const GemmArgs& g, int mt, int splits, int epi);
#ifdef GC_STAMPS
hipError_t set_gemm_ws_stamp_buffer(unsigned long long* p);     // diagnostic builds: 10 words per wave
#endif

// Both feed-forward layers in one launch (gc_ffw_fused): slab[z] = gelu(a @ W1[:, Fz] + b1[Fz]) @ W2[Fz, :]
// for hidden slices Fz of 256 columns; the hidden activations never leave LDS.  f16x3, WF16 weights.
struct FfwArgs {
  const float* a;      // [rows][d] float32 (the normed + conditioned residual stream)
  int rows, d, f;      // d % 128 == 0 (<= 512), f % 256 == 0
  const float* w1f;    // WF16 image of W1^T [f][d]
  const float* b1;     // [f]
  const float* w2f;    // WF16 image of W2^T [d][f]
  float* out;          // [f/256][rows][d] partial sums, one slab per hidden slice
  int round16;         // fp16-feature mode: the hidden activation is rounded to fp16
  int wt;              // 1: the slabs leave as write-through (sc1) stores (A/B switch GC_TUNE_WT_STORES & 1)
  // diagnostic builds only (-DGC_STAMPS, tools/stamp_ffw.cpp): 8 s_memtime stamps per wave, or nullptr
  unsigned long long* stamps;
};
hipError_t launch_ffw_fused(hipStream_t s, const FfwArgs& g);

// Row pass fused behind a full-width projection (gc_gemm_rowop): after y = A @ W (n == K == d_model,
// no K split), x <- x + bias + y and h <- cond(LayerNorm(x)), i.e. gc_rowop with one slab.
struct RowFuse {
  float* x;            // [rows][n] residual stream, updated in place
  const float* bias;   // [n] or nullptr
  const float* cond;   // per-batch [scale(n) | offset(n)]
  int cond_stride;
  int B;
  float* h;            // [rows][n]
  int round16;         // fp16-feature mode: x and h are rounded to fp16 when stored
};
// g as for launch_gemm_ws (WF16 weights; optional attention partials as A); g.out is unused.
hipError_t launch_gemm_rowop(hipStream_t s, int cls, const GemmArgs& g, const RowFuse& f);
#ifdef GC_STAMPS
hipError_t set_gemm_rowop_stamp_buffer(unsigned long long* p);   // diagnostic builds: 8 words per wave
#endif

// x += bias + sum of slabs (in place; skipped when both absent); h = cond(LN(x)) when h != nullptr
hipError_t launch_rowop(hipStream_t s, float* x, const float* bias, const float* partials, int n_slabs,
                        int rows, int d, int B, const float* cond, int cond_stride, float* h, bool h_s16,
                        bool round16 = false);

// S == 1: writes o directly; S > 1: writes partial (m, l, O) per key split for launch_attn_combine
hipError_t launch_attention(hipStream_t s, const float* qkv, float* o, float* part_o, float* part_ml,
                            int M, int B, int D, int H, int S, bool out_s16,
                            const int* tile_chunk_start, const int* union_idx, const unsigned* mask_bits,
                            int n_tiles, bool f16 = false, int max_chunks_per_tile = 0, bool feat16 = false);
// Attention on pre-split K / V planes (kv16 as written by launch_gemm_ws epi 3); q from qkv (f32).
// Always writes per-split partials when S > 1, o when S == 1 (as launch_attention).
hipError_t launch_attention_v2(hipStream_t s, const float* qkv, const void* kv16, float* o, float* part_o,
                               float* part_ml, int M, int B, int D, int H, int S, const int* tile_chunk_start,
                               const int* union_idx, const unsigned* mask_bits, int n_tiles,
                               int max_chunks_per_tile, bool feat16);
#ifdef GC_STAMPS
hipError_t set_attention_stamp_buffer(unsigned long long* p);   // diagnostic builds: 12 words per wave
#endif
hipError_t launch_attn_combine(hipStream_t s, const float* part_o, const float* part_ml, int M, int B,
                               int D, int H, int S, float* o, bool out_s16, bool round16 = false);

// grid input packing: xp[rows][kp] = [struct(3) | feats(c_in) | 0...]
hipError_t launch_pack_full(hipStream_t s, const float* grid_struct, const float* feats, int G, int B,
                            int c_in, int kp, float* xp);
// xp[row][3 + slot[c]] = scale * x[row][c]
hipError_t launch_write_noisy(hipStream_t s, const float* x, const int* slots, int rows, int c_out,
                              int kp, float scale, float* xp);
// out[item*B+b][c] = src[item][c] * scale[b][c] + offset[b][c]   (statically embedded latents)
hipError_t launch_affine_rows(hipStream_t s, const float* src, const float* cond, int cond_stride,
                              int items, int B, int w, float* out, bool round16 = false);
// f16x3 domain guard: *counter += (number of workgroups that saw a NaN / Inf in p[0..n)); p 16-byte aligned
hipError_t launch_finite_check(hipStream_t s, const float* p, size_t n, unsigned* counter);
// dst = a * src
hipError_t launch_scale(hipStream_t s, const float* src, float a, size_t n, float* dst);
// Per-channel context update of the packed conditioning (gc_rollout_advance; kinds in gencast_hip.h).
hipError_t launch_rollout_advance(hipStream_t s, const float* old_feats, const float* sample, const float* forcings,
                                  const int* kind, const int* src, const int* sidx, const float* a, const float* b,
                                  int rows, int c_in, int c_out, int n_forcing, float* new_feats);
// den = c_out*y + c_skip*x ; mid = a_mid*x + (1-a_mid)*den
hipError_t launch_dpm_first(hipStream_t s, const float* y, const float* x, float c_out, float c_skip,
                            float a_mid, size_t n, float* den, float* mid);
// md = c_out*y + c_skip*xmid ; x = a_next*x + (1-a_next)*md
hipError_t launch_dpm_second(hipStream_t s, const float* y, const float* xmid, float c_out,
                             float c_skip, float a_next, size_t n, float* x);

// Spherical white noise (gc_noise.hip): `count` N(0,1) values from Philox4x32-10 (key, stream), and the
// two-step synthesis out = (base ? base : 0) + scale * field, field [n_lat * n_lon][N] from coef [2][L][L][N].
hipError_t launch_noise_normals(hipStream_t s, float* out, size_t count, unsigned long long key,
                                unsigned long long stream);
hipError_t launch_noise_synthesis(hipStream_t s, const float* leg, const float* ctab, const float* stab,
                                  const float* coef, float* f, int L, int n_lat, int n_lon, int N,
                                  const float* base, float scale, float* out);

const char* kernel_class_name(int cls);

}  // namespace gc
