// Micro-benchmark harness for the transformer-side kernels at the nano shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gencast-flax-nnx_amd/csrc tools/bench_kernels.cpp \
//         gencast-flax-nnx_amd/csrc/gc_kernels.hip -o tools/bench_kernels
// Prints average kernel time over many back-to-back launches (hipEvents around the batch).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <cstdint>
#include <vector>

#include "gc_kernels.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// random S16-style payload: every 16-bit half is a normal fp16 in [2^-4, 2) with random sign
static float* dev_rand_s16(size_t n) {
  std::vector<uint16_t> h(2 * n);
  for (size_t i = 0; i < 2 * n; ++i) h[i] = (uint16_t)(((rand() & 1) << 15) | ((11 + rand() % 5) << 10) | (rand() & 0x3FF));
  float* d;
  CK(hipMalloc(&d, n * sizeof(float)));
  CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
  return d;
}

static float* dev_rand(size_t n, float scale = 1.0f) {
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = scale * ((float)rand() / RAND_MAX * 2.f - 1.f);
  float* d;
  CK(hipMalloc(&d, n * sizeof(float)));
  CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
  return d;
}

template <typename F>
static float time_it(hipStream_t s, int iters, F&& f) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) CK(f());
  CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) CK(f());
  CK(hipEventRecord(e1, s));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / iters;
}

static int g_af32 = 0;
int main(int argc, char** argv) {
  const int M = getenv("BK_M") ? atoi(getenv("BK_M")) : 2562, D = 256, F = 2048;
  hipStream_t s; CK(hipStreamCreate(&s));
  float* h = dev_rand((size_t)M * D);
  float* u = dev_rand((size_t)M * F);
  float* wqkv = dev_rand((size_t)3 * D * D, 0.06f);
  float* w1 = dev_rand((size_t)F * D, 0.06f);
  float* w2 = dev_rand((size_t)D * F, 0.02f);
  float* b1 = dev_rand(F);
  float* out = dev_rand((size_t)M * F);
  float* part = dev_rand((size_t)16 * M * D);
  const int iters = (argc > 3) ? atoi(argv[3]) : 200;
  const bool f16 = getenv("BK_F16") != nullptr;
  if (f16) { h = dev_rand_s16((size_t)M * D); u = dev_rand_s16((size_t)M * F); wqkv = dev_rand_s16((size_t)3 * D * D);
             w1 = dev_rand_s16((size_t)F * D); w2 = dev_rand_s16((size_t)D * F); }
  auto gemm = [&](const char* name, int cls, const float* a, int lda, const float* wt, int ldw, int n, int k,
                  int splits, const float* bias, int act, float* o, int ldo, int mt, int epi) {
    g_af32 = getenv("BK_AF32") ? 1 : 0;
    gc::GemmArgs g{};
    g.a = a; g.lda = lda; g.a_f32 = g_af32; g.wt = wt; g.ldw = ldw; g.rows = M; g.n = n; g.k_slice = k / splits;
    g.bias = bias; g.act = act; g.out = o; g.ldo = ldo;
    float us = time_it(s, iters, [&] { return gc::launch_gemm(s, cls, g, mt, splits, epi, f16); });
    double fl = 2.0 * M * n * k;
    printf("%-34s mt=%d splits=%d  %8.2f us  %6.1f TF/s\n", name, mt, splits, us, fl / us * 1e-6);
  };
  if (argc <= 1) {  // attention with synthetic tiles: 81 tiles x 13 chunks, ~50 % mask density, clustered keys
    const int T = (M + 31) / 32, CH = 13, H = 4;
    std::vector<int> tstart(T + 1), uni((size_t)T * CH * 32);
    std::vector<unsigned> mask((size_t)T * CH * 32);
    for (int t = 0; t <= T; ++t) tstart[t] = t * CH;
    for (int t = 0; t < T; ++t)
      for (int i = 0; i < CH * 32; ++i) {
        int k = t * 32 - 190 + i + (rand() % 5);
        uni[(size_t)t * CH * 32 + i] = ((k % M) + M) % M;
        mask[(size_t)t * CH * 32 + i] = (unsigned)rand() ^ ((unsigned)rand() << 16);
      }
    int *d_ts, *d_un; unsigned* d_mk;
    CK(hipMalloc(&d_ts, tstart.size() * 4)); CK(hipMemcpy(d_ts, tstart.data(), tstart.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_un, uni.size() * 4)); CK(hipMemcpy(d_un, uni.data(), uni.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_mk, mask.size() * 4)); CK(hipMemcpy(d_mk, mask.data(), mask.size() * 4, hipMemcpyHostToDevice));
    float* qkv = dev_rand((size_t)M * 3 * D);
    float* att = dev_rand((size_t)M * D);
    float* po = dev_rand((size_t)T * 8 * H * 32 * 64);
    float* pml = dev_rand((size_t)T * 8 * H * 32 * 2);
    for (int S : {1, 2, 3, 4, 6, 8}) {
      float us = time_it(s, iters, [&] { return gc::launch_attention(s, qkv, att, po, pml, M, 1, D, H, S, false, d_ts, d_un, d_mk, T); });
      float us2 = S > 1 ? time_it(s, iters, [&] { return gc::launch_attn_combine(s, po, pml, M, 1, D, H, S, att, false); }) : 0.f;
      double fl = 4.0 * T * CH * 32 * 32 * D;   // dense tile flops actually executed
      printf("attention S=%d  %8.2f us (+combine %5.2f us)  %6.1f TF/s executed\n", S, us, us2, fl / us * 1e-6);
    }
  }
  const char* only = argc > 1 ? argv[1] : nullptr;
  if (only && std::string(only) == "ws") {   // weight-streaming f16x3 GEMM at the four nano shapes
    float* hf = dev_rand((size_t)M * D);
    float* uf = dev_rand((size_t)M * F);
    float* wq = dev_rand_s16((size_t)3 * D * D);
    float* wa = dev_rand_s16((size_t)F * D);
    float* wb = dev_rand_s16((size_t)D * F);
    auto ws = [&](const char* name, int cls, const float* a, int lda, const float* wt, int n, int k, int splits,
                  const float* bias, int act, float* o, int ldo, int mt, int epi) {
      gc::GemmArgs g{};
      g.a = a; g.lda = lda; g.a_f32 = 1; g.wt = wt; g.ldw = k; g.rows = M; g.n = n; g.k_slice = k / splits;
      g.bias = bias; g.act = act; g.out = o; g.ldo = ldo;
      float us = time_it(s, iters, [&] { return gc::launch_gemm_ws(s, cls, g, mt, splits, epi); });
      printf("ws %-30s mt=%d splits=%d  %8.2f us  %6.1f TF/s\n", name, mt, splits, us, 2.0 * M * n * k / us * 1e-6);
    };
    for (int mt = 1; mt <= 2; ++mt) {
      ws("qkv  [Mx256]x[256x768]", gc::KC_GEMM_QKV, hf, D, wq, 3 * D, D, 1, nullptr, 0, out, 3 * D, mt, 0);
      ws("ffw1 [Mx256]x[256x2048]+gelu", gc::KC_GEMM_FFW1, hf, D, wa, F, D, 1, b1, 1, out, F, mt, 0);
      for (int sp : {2, 4, 8}) ws("ffw2 [Mx2048]x[2048x256]", gc::KC_GEMM_FFW2, uf, F, wb, D, F, sp, nullptr, 0, part, D, mt, 1);
      for (int sp : {1, 2}) ws("out  [Mx256]x[256x256]", gc::KC_GEMM_OUT, hf, D, wq, D, D, sp, nullptr, 0, part, D, mt, 1);
    }
    return 0;
  }
  if (only) {   // single-config mode for rocprofv3 counter runs: bench_kernels ffw1 <mt> [iters]
    const int mt = argc > 2 ? atoi(argv[2]) : 1;
    if (std::string(only) == "floor") {
      for (int kk : {32, 64, 128, 256}) {
        char nm[64]; snprintf(nm, 64, "ffw1-shape K=%d (no gelu)", kk);
        gemm(nm, gc::KC_GEMM_FFW1, h, D, w1, D, F, kk, 1, b1, 0, out, F, mt, 0);
        snprintf(nm, 64, "ffw1-shape K=%d slabs", kk);
        gemm(nm, gc::KC_GEMM_FFW1, h, D, w1, D, F, kk, 1, nullptr, 0, out, F, mt, 1);
        snprintf(nm, 64, "qkv-shape K=%d", kk);
        gemm(nm, gc::KC_GEMM_QKV, h, D, wqkv, D, 3 * D, kk, 1, nullptr, 0, out, 3 * D, mt, 0);
        snprintf(nm, 64, "out-shape K=%d", kk);
        gemm(nm, gc::KC_GEMM_OUT, h, D, wqkv, D, D, kk, 1, nullptr, 0, out, D, mt, 1);
      }
    }
    if (std::string(only) == "ffw1") gemm("ffw1 (no gelu)", gc::KC_GEMM_FFW1, h, D, w1, D, F, D, 1, b1, 0, out, F, mt, 0);
    if (std::string(only) == "ffw2") gemm("ffw2 splits 4", gc::KC_GEMM_FFW2, u, F, w2, F, D, F, 4, nullptr, 0, part, D, mt, 1);
    return 0;
  }
  for (int mt = 1; mt <= 2; ++mt) {
    gemm("qkv   [2562x256]x[256x768]", gc::KC_GEMM_QKV, h, D, wqkv, D, 3 * D, D, 1, nullptr, 0, out, 3 * D, mt, 0);
    gemm("ffw1  [2562x256]x[256x2048] +gelu", gc::KC_GEMM_FFW1, h, D, w1, D, F, D, 1, b1, 1, out, F, mt, 0);
    gemm("ffw1  (no gelu)", gc::KC_GEMM_FFW1, h, D, w1, D, F, D, 1, b1, 0, out, F, mt, 0);
    for (int sp : {1, 2, 4, 8})
      gemm("ffw2  [2562x2048]x[2048x256]", gc::KC_GEMM_FFW2, u, F, w2, F, D, F, sp, nullptr, 0, part, D, mt, 1);
    for (int sp : {1, 2})
      gemm("out   [2562x256]x[256x256]", gc::KC_GEMM_OUT, h, D, wqkv, D, D, D, sp, nullptr, 0, part, D, mt, 1);
  }
  return 0;
}
