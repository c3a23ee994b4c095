// In-kernel timeline of the weight-streaming GEMM (gc_gemm_ws_kernel) at a chosen shape (diagnostic build).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGC_STAMPS -I gencast-flax-nnx_amd/csrc tools/stamp_gemm_ws.cpp \
//         gencast-flax-nnx_amd/csrc/gc_kernels.hip -o tools/stamp_gemm_ws
//   tools/stamp_gemm_ws [rows K N mt epi act [launches]] | [launches]     default: the 1-degree FFW layer 1 (10242 512 2048 2 0 1), 20 launches
// Prints, per phase, the median / p90 over waves of the s_memtime deltas (shader cycles) of the last of
// several back-to-back launches, the wave lifetime, and the launch's wall time from HIP events.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gc_kernels.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

static float* dev_rand(size_t n, float scale = 1.0f) {
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = scale * ((float)rand() / RAND_MAX * 2.f - 1.f);
  float* d;
  CK(hipMalloc(&d, n * sizeof(float)));
  CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
  return d;
}
static float* dev_rand_f16pairs(size_t n) {   // WF16-like payload: every half a normal fp16 of magnitude < 1
  std::vector<uint16_t> h(2 * n);
  for (size_t i = 0; i < 2 * n; ++i) h[i] = (uint16_t)(((rand() & 1) << 15) | ((9 + rand() % 5) << 10) | (rand() & 0x3FF));
  float* d;
  CK(hipMalloc(&d, n * sizeof(float)));
  CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
  return d;
}

int main(int argc, char** argv) {
  int rows = 10242, K = 512, N = 2048, mt = 2, epi = 0, act = 1;
  if (argc >= 7) { rows = atoi(argv[1]); K = atoi(argv[2]); N = atoi(argv[3]); mt = atoi(argv[4]); epi = atoi(argv[5]); act = atoi(argv[6]); }
  hipStream_t s; CK(hipStreamCreate(&s));
  gc::GemmArgs g{};
  g.a = dev_rand((size_t)rows * K); g.lda = K; g.a_f32 = 1;
  g.wt = dev_rand_f16pairs((size_t)N * K); g.ldw = K;
  g.rows = rows; g.n = N; g.k_slice = K; g.bias = (epi == 0) ? dev_rand(N) : nullptr; g.act = act;
  float* out; CK(hipMalloc(&out, (size_t)rows * N * sizeof(float))); g.out = out; g.ldo = N;
  void* kv = nullptr;
  if (epi == 3) { CK(hipMalloc(&kv, (size_t)rows * 4 * (N / 3) * 2)); g.kv16 = kv; g.kv_d = N / 3; }
  const int cls = epi == 3 ? gc::KC_GEMM_QKV : gc::KC_GEMM_FFW1;
  const int BM = 32 * mt;
  const int wgs = (((rows + BM - 1) / BM) * (N / 128) + 7) / 8 * 8, waves = 4;
  unsigned long long* st; CK(hipMalloc(&st, (size_t)wgs * waves * 10 * sizeof(unsigned long long)));
  CK(hipMemset(st, 0, (size_t)wgs * waves * 10 * sizeof(unsigned long long)));
  CK(gc::set_gemm_ws_stamp_buffer(st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) CK(gc::launch_gemm_ws(s, cls, g, mt, 1, epi));
  CK(hipEventRecord(e0, s));
  const int reps = argc >= 8 ? atoi(argv[7]) : (argc == 2 ? atoi(argv[1]) : 20);
  for (int i = 0; i < reps; ++i) CK(gc::launch_gemm_ws(s, cls, g, mt, 1, epi));
  CK(hipEventRecord(e1, s));
  CK(hipStreamSynchronize(s));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = 1e3 * ms / reps, tf = 2.0 * rows * K * N / (us * 1e-6) / 1e12;
  printf("rows %d K %d N %d mt %d epi %d act %d: %.1f us per launch = %.1f TFLOP/s algorithmic (%.3f of the f16x3 ceiling 838.9)\n",
         rows, K, N, mt, epi, act, us, tf, tf / 838.9);
  std::vector<unsigned long long> h((size_t)wgs * waves * 10);
  CK(hipMemcpy(h.data(), st, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  std::vector<size_t> live;
  for (size_t w = 0; w < (size_t)wgs * waves; ++w) if (h[w * 10]) live.push_back(w);
  auto stat = [&](const char* name, auto f) {
    std::vector<unsigned long long> d;
    for (size_t w : live) d.push_back(f(&h[w * 10]));
    std::sort(d.begin(), d.end());
    printf("%-44s median %7llu  p10 %7llu  p90 %7llu  max %7llu\n", name, d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10], d.back());
  };
  printf("waves %zu (4 per workgroup)\n", live.size());
  stat("prologue: W ring + first A chunk issued", [](const unsigned long long* p) { return p[1] - p[0]; });
  stat("(sum) barrier: previous chunk released", [](const unsigned long long* p) { return p[2]; });
  stat("(sum) A wait + split + LDS writes", [](const unsigned long long* p) { return p[3]; });
  stat("(sum) barrier: chunk staged", [](const unsigned long long* p) { return p[4]; });
  stat("(sum) product loops", [](const unsigned long long* p) { return p[5]; });
  stat("epilogue: stores issued", [&](const unsigned long long* p) { return p[6] - (p[1] + p[2] + p[3] + p[4] + p[5]); });
  stat("store drain", [](const unsigned long long* p) { return p[7] - p[6]; });
  stat("wave lifetime", [](const unsigned long long* p) { return p[7] - p[0]; });
  // s_memrealtime is one chip-wide 100 MHz counter: launch span and the shader clock the waves saw
  unsigned long long r_first = ~0ull, r_last = 0;
  double clk = 0;
  for (size_t w : live) {
    r_first = std::min(r_first, h[w * 10 + 8]); r_last = std::max(r_last, h[w * 10 + 9]);
    clk += (double)(h[w * 10 + 7] - h[w * 10]) / (double)std::max<unsigned long long>(h[w * 10 + 9] - h[w * 10 + 8], 1) * 100.0;
  }
  clk /= live.size();
  double busy = 0;
  for (size_t w : live) busy += (double)(h[w * 10 + 9] - h[w * 10 + 8]);
  printf("launch span first entry -> last drain: %.1f us; shader clock seen by the waves %.0f MHz; average waves resident %.0f "
         "(= %.2f workgroups per CU); MFMA floor per wave = %d cycles (k16 steps x %d MFMAs x 32)\n",
         (r_last - r_first) / 100.0, clk, busy / (double)(r_last - r_first), busy / (double)(r_last - r_first) / 4.0 / 256.0,
         (K / 16) * 3 * mt * 32, 3 * mt);
  stat("wave start offset within the launch (us x100)", [&](const unsigned long long* p) { return p[8] - r_first; });
  return 0;
}
