// Correctness + timing harness of the large-tile GEMM (tools/gemm_lt/gc_gemm_lt.hip -- an EXPERIMENT kept with its
// evidence, not part of libgencast_hip.so since round 5: three generations, all parity-green, none faster end to end;
// DESIGN.md section 5c) at the 1-degree shapes, next to the weight-streaming kernel of the product, on one box and in
// one process (interleaved rounds).  Build from the repo root:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I tools/gemm_lt -I gencast-flax-nnx_amd/csrc tools/gemm_lt/bench_gemm_lt.cpp \
//         tools/gemm_lt/gc_gemm_lt.hip gencast-flax-nnx_amd/csrc/gc_kernels.hip -o tools/bench_gemm_lt
// Checks sampled outputs of every epilogue against a float64 reference of the same (float32) inputs.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "gc_gemm_lt.h"
#include "gc_kernels.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

static uint16_t f2h(float f) {
  _Float16 h = (_Float16)f;
  uint16_t b;
  memcpy(&b, &h, 2);
  return b;
}
static float h2f(uint16_t b) {
  _Float16 h;
  memcpy(&h, &b, 2);
  return (float)h;
}
static int kpos(int kk, bool perm) {      // position (hk, i) -> lane half, element of k16-local index kk
  // natural: kk = 8 hk + i; permuted: kk = 8 (i >> 2) + 4 hk + (i & 3)
  if (!perm) return ((kk >> 3) << 3) | (kk & 7);
  const int hk = (kk >> 2) & 1, i = ((kk >> 3) << 2) | (kk & 3);
  return (hk << 3) | i;
}
// [rows][k] f32 -> fragment-order image; tiles = row tiles allocated (>= ceil(rows / 32)); planes 2 (hi, lo) or 1
static std::vector<uint16_t> encode_frag(const std::vector<float>& m, int rows, int k, int tiles, int planes, bool perm) {
  const size_t steps = k / 16;
  std::vector<uint16_t> o((size_t)tiles * steps * planes * 512, 0);
  for (int r = 0; r < rows; ++r)
    for (int kk = 0; kk < k; ++kk) {
      const float x = m[(size_t)r * k + kk];
      const uint16_t hi = f2h(x);
      const uint16_t lo = f2h((x - h2f(hi)) * 2048.0f);
      const int p = kpos(kk % 16, perm), hk = p >> 3, i = p & 7;
      const size_t blk = ((size_t)(r / 32) * steps + kk / 16) * planes;
      const size_t lane = (size_t)hk * 32 + r % 32;
      o[(blk * 64 + lane) * 8 + i] = hi;
      if (planes == 2) o[((blk + 1) * 64 + lane) * 8 + i] = lo;
    }
  return o;
}
template <typename T>
static T* to_dev(const std::vector<T>& v) {
  T* d;
  CK(hipMalloc(&d, v.size() * sizeof(T)));
  CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}
static std::vector<float> rnd(size_t n, float scale, uint32_t seed) {
  std::mt19937 g(seed);
  std::normal_distribution<float> d(0.f, scale);
  std::vector<float> v(n);
  for (auto& x : v) x = d(g);
  return v;
}
static double gelu(double x) { return 0.5 * x * (1.0 + std::tanh(0.7978845608028654 * (x + 0.044715 * x * x * x))); }

template <typename F>
static float time_us(hipStream_t s, int iters, F&& f) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) CK(f());
  CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) CK(f());
  CK(hipEventRecord(e1, s));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / iters;
}

int main(int argc, char** argv) {
  const int M = getenv("LT_M") ? atoi(getenv("LT_M")) : 10242, D = 512, F = 2048;
  const int iters = getenv("LT_ITERS") ? atoi(getenv("LT_ITERS")) : 200;
  const int gx = getenv("LT_GX") ? atoi(getenv("LT_GX")) : 0;
  const int shape = getenv("LT_SHAPE") ? atoi(getenv("LT_SHAPE")) : 0;
  hipStream_t s;
  CK(hipStreamCreate(&s));
  const int tiles = gc_lt::lt_row_tiles(M);
  auto h_h = rnd((size_t)M * D, 1.0f, 1), w1 = rnd((size_t)F * D, 0.05f, 2), w2 = rnd((size_t)D * F, 0.03f, 3),
       wq = rnd((size_t)3 * D * D, 0.05f, 4), b1 = rnd(F, 0.3f, 5), u_h = rnd((size_t)M * F, 0.7f, 6);
  int bad = 0;
  if (getenv("LT_CONST")) {                      // debugging: constant operands, every pre-activation = K / 64
    for (auto& x : h_h) x = 1.f;
    for (auto& x : u_h) x = 1.f;
    for (auto& x : w1) x = 1.f / 64;
    for (auto& x : w2) x = 1.f / 64;
    for (auto& x : wq) x = 1.f / 64;
    for (auto& x : b1) x = 0.f;
    const int am = getenv("LT_AMASK") ? atoi(getenv("LT_AMASK")) : 15, wm = getenv("LT_WMASK") ? atoi(getenv("LT_WMASK")) : 15;
    for (size_t i = 0; i < h_h.size(); ++i) if (!((am >> ((i % D) / 16 % 4)) & 1)) h_h[i] = 0.f;
    for (size_t i = 0; i < w1.size(); ++i) if (!((wm >> ((i % D) / 16 % 4)) & 1)) w1[i] = 0.f;
    for (size_t i = 0; i < wq.size(); ++i) if (!((wm >> ((i % D) / 16 % 4)) & 1)) wq[i] = 0.f;
  }
  for (int a16 = 0; a16 < 2; ++a16) {
    const int PA = a16 ? 1 : 2;
    std::vector<float> hh = h_h, uu = u_h;
    if (a16) {                                   // exact-fp16 activations
      for (auto& x : hh) x = h2f(f2h(x));
      for (auto& x : uu) x = h2f(f2h(x));
    }
    uint16_t* d_h = to_dev(encode_frag(hh, M, D, tiles, PA, false));
    uint16_t* d_u = to_dev(encode_frag(uu, M, F, tiles, PA, true));
    uint16_t* d_w1 = to_dev(encode_frag(w1, F, D, F / 32, 2, false));
    uint16_t* d_w2 = to_dev(encode_frag(w2, D, F, D / 32, 2, true));
    uint16_t* d_wq = to_dev(encode_frag(wq, 3 * D, D, 3 * D / 32, 2, false));
    float* d_b1 = to_dev(b1);
    uint16_t* d_hid;                             // FFW-1 output image
    CK(hipMalloc(&d_hid, (size_t)tiles * (F / 16) * PA * 1024));
    CK(hipMemset(d_hid, 0, (size_t)tiles * (F / 16) * PA * 1024));
    float* d_slab;
    CK(hipMalloc(&d_slab, (size_t)4 * M * D * 4));
    float* d_q;
    CK(hipMalloc(&d_q, (size_t)M * D * 4));
    uint16_t* d_kv;
    CK(hipMalloc(&d_kv, (size_t)M * 4 * D * 2));
    CK(hipMemset(d_kv, 0, (size_t)M * 4 * D * 2));

    gc_lt::LtArgs f1{};
    f1.a = d_h; f1.a_steps = D / 16; f1.wt = (const float*)d_w1; f1.w_steps = D / 16; f1.rows = M; f1.n = F;
    f1.k_steps = D / 16; f1.splits = 1; f1.bias = d_b1; f1.act = 1; f1.out = (float*)d_hid; f1.out_steps = F / 16;
    f1.round16 = a16; f1.gx = gx; f1.shape = shape; f1.dbg = getenv("LT_DBG") ? atoi(getenv("LT_DBG")) : 0;
    gc_lt::LtArgs f2{};
    const int sp2 = getenv("LT_SPLITS") ? atoi(getenv("LT_SPLITS")) : 4;
    f2.a = d_u; f2.a_steps = F / 16; f2.wt = (const float*)d_w2; f2.w_steps = F / 16; f2.rows = M; f2.n = D;
    f2.k_steps = F / 16 / sp2; f2.splits = sp2; f2.out = d_slab; f2.ldo = D; f2.gx = gx; f2.shape = shape;
    gc_lt::LtArgs q{};
    q.a = d_h; q.a_steps = D / 16; q.wt = (const float*)d_wq; q.w_steps = D / 16; q.rows = M; q.n = 3 * D;
    q.k_steps = D / 16; q.splits = 1; q.out = d_q; q.ldo = D; q.kv16 = d_kv; q.kv_d = D; q.round16 = a16; q.gx = gx; q.shape = shape;

    if (shape == 16 && !a16) {                  // in-kernel stamps of the FFW-1 launch, after a warm-up of 300 launches
      gc_lt::LtArgs fs = f1;
      fs.shape = 6;
      for (int i = 0; i < 300; ++i) CK(gc_lt::launch_gemm_lt(s, gc::KC_GEMM_FFW1, fs, gc_lt::LT_EPI_AF16, false));
      unsigned long long* d_st;
      const int nwg = 8 * 4096;
      CK(hipMalloc(&d_st, (size_t)nwg * 8 * 8));
      CK(hipMemset(d_st, 0, (size_t)nwg * 8 * 8));
      fs.shape = 16; fs.stamps = d_st;
      CK(gc_lt::launch_gemm_lt(s, gc::KC_GEMM_FFW1, fs, gc_lt::LT_EPI_AF16, false));
      CK(hipStreamSynchronize(s));
      std::vector<unsigned long long> st((size_t)nwg * 8);
      CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
      double tot = 0, bar = 0, body = 0, rtk = 0, epi = 0, tiles = 0, tmax = 0; int n = 0;
      for (int w = 0; w < nwg; ++w) if (st[w * 8]) { tot += st[w * 8]; bar += st[w * 8 + 1]; body += st[w * 8 + 2]; rtk += st[w * 8 + 3]; epi += st[w * 8 + 4]; tiles += st[w * 8 + 5]; tmax = std::fmax(tmax, (double)st[w * 8 + 3]); ++n; }
      printf("stamps over %d workgroups (%.0f tiles): consumer lifetime %.0f cycles = %.2f us (longest %.2f us), clock %.0f MHz; per tile: barrier waits %.0f, stage bodies %.0f, epilogue %.0f cycles\n",
             n, tiles, tot / n, rtk / n / 100.0, tmax / 100.0, tot / rtk * 100.0, bar / tiles, body / tiles, epi / tiles);
      return 0;
    }
    CK(gc_lt::launch_gemm_lt(s, gc::KC_GEMM_FFW1, f1, gc_lt::LT_EPI_AF16, a16));
    CK(gc_lt::launch_gemm_lt(s, gc::KC_GEMM_FFW2, f2, gc_lt::LT_EPI_F32, a16));
    CK(gc_lt::launch_gemm_lt(s, gc::KC_GEMM_QKV, q, gc_lt::LT_EPI_QKV, a16));
    CK(hipStreamSynchronize(s));

    // ---- checks on sampled outputs
    std::vector<uint16_t> hid((size_t)tiles * (F / 16) * PA * 512);
    CK(hipMemcpy(hid.data(), d_hid, hid.size() * 2, hipMemcpyDeviceToHost));
    std::vector<float> slab((size_t)sp2 * M * D), qo((size_t)M * D);
    CK(hipMemcpy(slab.data(), d_slab, slab.size() * 4, hipMemcpyDeviceToHost));
    std::vector<uint16_t> kv((size_t)M * 4 * D), q16((size_t)M * D);
    CK(hipMemcpy(kv.data(), d_kv, kv.size() * 2, hipMemcpyDeviceToHost));
    if (a16) CK(hipMemcpy(q16.data(), d_q, q16.size() * 2, hipMemcpyDeviceToHost));
    else CK(hipMemcpy(qo.data(), d_q, qo.size() * 4, hipMemcpyDeviceToHost));
    std::mt19937 g(7);
    double e1 = 0, e2 = 0, eq = 0;
    const int rows_probe[] = {0, 1, 31, 32, 63, 64, 127, 128, M / 2, M - 3, M - 2, M - 1};
    for (int t = 0; t < 3000; ++t) {
      const int r = t < 12 * 40 ? rows_probe[t % 12] : (int)(g() % M);
      {  // FFW-1: hidden[r][c]
        const int c = g() % F;
        double ref = b1[c];
        for (int k = 0; k < D; ++k) ref += (double)hh[(size_t)r * D + k] * w1[(size_t)c * D + k];
        ref = gelu(ref);
        const int p = kpos(c % 16, true), hk = p >> 3, i = p & 7;
        const size_t blk = ((size_t)(r / 32) * (F / 16) + c / 16) * PA;
        const size_t lane = (size_t)hk * 32 + r % 32;
        double got = h2f(hid[(blk * 64 + lane) * 8 + i]);
        if (!a16) got += h2f(hid[((blk + 1) * 64 + lane) * 8 + i]) / 2048.0;
        const double tol = a16 ? 1.5e-3 * (1 + std::fabs(ref)) : 2e-5;
        if (!(std::fabs(got - ref) <= tol)) { if (bad++ < 10) printf("FFW1 a16=%d r=%d c=%d got %g ref %g\n", a16, r, c, got, ref); }
        e1 = std::fmax(e1, std::fabs(got - ref));
      }
      {  // FFW-2: sum of slabs [r][c]
        const int c = g() % D;
        double ref = 0;
        for (int k = 0; k < F; ++k) ref += (double)uu[(size_t)r * F + k] * w2[(size_t)c * F + k];
        double got = 0;
        for (int z = 0; z < sp2; ++z) got += slab[((size_t)z * M + r) * D + c];
        if (!(std::fabs(got - ref) <= 3e-5)) { if (bad++ < 10) printf("FFW2 a16=%d r=%d c=%d got %g ref %g\n", a16, r, c, got, ref); }
        e2 = std::fmax(e2, std::fabs(got - ref));
      }
      {  // QKV
        const int c = g() % (3 * D);
        double ref = 0;
        for (int k = 0; k < D; ++k) ref += (double)hh[(size_t)r * D + k] * wq[(size_t)c * D + k];
        double got;
        if (c < D) got = a16 ? h2f(q16[(size_t)r * D + c]) : qo[(size_t)r * D + c];
        else {
          const int w = c / D - 1, ci = c % D;
          got = h2f(kv[(size_t)r * 4 * D + w * 2 * D + ci]);
          if (!a16) got += h2f(kv[(size_t)r * 4 * D + w * 2 * D + D + ci]) / 2048.0;
        }
        const double tol = a16 ? 1.5e-3 * (1 + std::fabs(ref)) : 2e-5;
        if (!(std::fabs(got - ref) <= tol)) { if (bad++ < 10) printf("QKV a16=%d r=%d c=%d got %g ref %g\n", a16, r, c, got, ref); }
        eq = std::fmax(eq, std::fabs(got - ref));
      }
    }
    if (getenv("LT_CONST")) {
      for (int r : {0, 1, 33, 64, 100, 127, 128, 200})
        printf("r=%d slab0 c0..3: %g %g %g %g  c=127 %g c=128 %g c=300 %g | q: %g %g\n", r, slab[(size_t)r * D], slab[(size_t)r * D + 1],
               slab[(size_t)r * D + 2], slab[(size_t)r * D + 3], slab[(size_t)r * D + 127], slab[(size_t)r * D + 128],
               slab[(size_t)r * D + 300], qo[(size_t)r * D], qo[(size_t)r * D + 77]);
    }
    printf("a16=%d  max abs err: FFW1 %.3g  FFW2 %.3g  QKV %.3g  (bad %d)\n", a16, e1, e2, eq, bad);

    // ---- the weight-streaming kernels on the same shapes (their own layouts; values do not matter for time)
    float* d_hf = to_dev(hh);
    float* d_uf;
    CK(hipMalloc(&d_uf, (size_t)M * F * 4));
    CK(hipMemset(d_uf, 0, (size_t)M * F * 4));
    gc::GemmArgs w1a{};
    w1a.a = d_hf; w1a.lda = D; w1a.a_f32 = 1; w1a.wt = (const float*)d_w1; w1a.ldw = D; w1a.rows = M; w1a.n = F; w1a.k_slice = D;
    w1a.bias = d_b1; w1a.act = 1; w1a.out = d_uf; w1a.ldo = F;
    gc::GemmArgs w2a{};
    w2a.a = d_uf; w2a.lda = F; w2a.a_f32 = 1; w2a.wt = (const float*)d_w2; w2a.ldw = F; w2a.rows = M; w2a.n = D; w2a.k_slice = F;
    w2a.out = d_slab; w2a.ldo = D;
    gc::GemmArgs wqa{};
    wqa.a = d_hf; wqa.lda = D; wqa.a_f32 = 1; wqa.wt = (const float*)d_wq; wqa.ldw = D; wqa.rows = M; wqa.n = 3 * D; wqa.k_slice = D;
    wqa.out = d_q; wqa.ldo = 3 * D; wqa.kv16 = d_kv; wqa.kv_d = D;
    float* d_qkv3;
    CK(hipMalloc(&d_qkv3, (size_t)M * 3 * D * 4));
    wqa.out = d_qkv3;
    for (int round = 0; round < 3; ++round) {
      const float t1 = time_us(s, iters, [&] { return gc_lt::launch_gemm_lt(s, gc::KC_GEMM_FFW1, f1, gc_lt::LT_EPI_AF16, a16); });
      const float t2 = time_us(s, iters, [&] { return gc_lt::launch_gemm_lt(s, gc::KC_GEMM_FFW2, f2, gc_lt::LT_EPI_F32, a16); });
      const float t3 = time_us(s, iters, [&] { return gc_lt::launch_gemm_lt(s, gc::KC_GEMM_QKV, q, gc_lt::LT_EPI_QKV, a16); });
      float o1 = 0, o2 = 0, o3 = 0;
      if (!a16) {
        o1 = time_us(s, iters, [&] { return gc::launch_gemm_ws(s, gc::KC_GEMM_FFW1, w1a, 2, 1, 0); });
        o2 = time_us(s, iters, [&] { return gc::launch_gemm_ws(s, gc::KC_GEMM_FFW2, w2a, 2, 1, 1); });
        o3 = time_us(s, iters, [&] { return gc::launch_gemm_ws(s, gc::KC_GEMM_QKV, wqa, 2, 1, 3); });
      }
      const double gf1 = 2.0 * M * D * F * 1e-9, gfq = 2.0 * M * D * 3 * D * 1e-9;
      printf("a16=%d round %d  lt: FFW1 %.1f us (%.0f TF/s)  FFW2[s%d] %.1f us (%.0f)  QKV %.1f us (%.0f)   ws: FFW1 %.1f  FFW2 %.1f  QKV %.1f\n",
             a16, round, t1, gf1 / t1 * 1e3, sp2, t2, gf1 / t2 * 1e3, t3, gfq / t3 * 1e3, o1, o2, o3);
    }
    CK(hipFree(d_h)); CK(hipFree(d_u)); CK(hipFree(d_w1)); CK(hipFree(d_w2)); CK(hipFree(d_wq)); CK(hipFree(d_b1));
    CK(hipFree(d_hid)); CK(hipFree(d_slab)); CK(hipFree(d_q)); CK(hipFree(d_kv)); CK(hipFree(d_hf)); CK(hipFree(d_uf)); CK(hipFree(d_qkv3));
  }
  printf(bad ? "FAILED (%d bad)\n" : "OK\n", bad);
  return bad ? 1 : 0;
}
