// In-kernel timeline of the fused FFW kernel at the nano shape (diagnostic build: -DGC_STAMPS).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGC_STAMPS -I gencast-flax-nnx_amd/csrc tools/stamp_ffw.cpp \
//         gencast-flax-nnx_amd/csrc/gc_kernels.hip -o tools/stamp_ffw
// Prints, per phase, the median / max over waves of the s_memtime deltas (shader cycles) of the last of
// several back-to-back launches, and the kernel's span (first entry -> last drain).
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

int main() {
  const int M = 2562, D = 256, F = 2048;
  hipStream_t s; CK(hipStreamCreate(&s));
  gc::FfwArgs a{};
  a.a = dev_rand((size_t)M * D); a.rows = M; a.d = D; a.f = F;
  a.w1f = dev_rand_f16pairs((size_t)F * D); a.b1 = dev_rand(F); a.w2f = dev_rand_f16pairs((size_t)D * F);
  float* out; CK(hipMalloc(&out, (size_t)8 * M * D * sizeof(float))); a.out = out; a.round16 = 0;
  const int wgs = ((M + 95) / 96) * (F / 256), waves = 8;
  unsigned long long* st; CK(hipMalloc(&st, (size_t)wgs * waves * 8 * sizeof(unsigned long long)));
  a.stamps = st;
  for (int i = 0; i < 20; ++i) CK(gc::launch_ffw_fused(s, a));
  CK(hipStreamSynchronize(s));
  std::vector<unsigned long long> h((size_t)wgs * waves * 8);
  CK(hipMemcpy(h.data(), st, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  const char* names[7] = {"a tile: load+split+LDS", "barrier wait", "phase-1 products", "gelu + LDS + barrier",
                          "phase-2 products", "slab stores issue", "store drain"};
  unsigned long long t_first = ~0ull, t_last = 0;
  for (size_t w = 0; w < (size_t)wgs * waves; ++w) { t_first = std::min(t_first, h[w * 8]); t_last = std::max(t_last, h[w * 8 + 7]); }
  printf("workgroups %d, waves %d; span first entry -> last drain: %llu cycles\n", wgs, wgs * waves, t_last - t_first);
  for (int p = 0; p < 7; ++p) {
    std::vector<unsigned long long> d;
    for (size_t w = 0; w < (size_t)wgs * waves; ++w) d.push_back(h[w * 8 + p + 1] - h[w * 8 + p]);
    std::sort(d.begin(), d.end());
    printf("%-26s median %7llu  p10 %7llu  p90 %7llu  max %7llu cycles\n", names[p], d[d.size() / 2], d[d.size() / 10],
           d[d.size() * 9 / 10], d.back());
  }
  std::vector<unsigned long long> life, start;
  for (size_t w = 0; w < (size_t)wgs * waves; ++w) { life.push_back(h[w * 8 + 7] - h[w * 8]); start.push_back(h[w * 8] - t_first); }
  std::sort(life.begin(), life.end()); std::sort(start.begin(), start.end());
  printf("wave lifetime median %llu max %llu; wave start offset median %llu max %llu cycles (s_memtime: 100 MHz-independent shader clock)\n",
         life[life.size() / 2], life.back(), start[start.size() / 2], start.back());
  return 0;
}
