// In-kernel timeline of the out-projection + partial-merge + row-pass kernel (gc_gemm_rowop) at the nano shape.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGC_STAMPS -I gencast-flax-nnx_amd/csrc tools/stamp_rowop.cpp \
//         gencast-flax-nnx_amd/csrc/gc_kernels.hip -o tools/stamp_rowop
//   tools/stamp_rowop [rows D heads splits launches]        default 2562 256 4 3 2000
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gc_kernels.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

static float* dev_rand(size_t n, float lo = -1.f, float hi = 1.f) {
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = lo + (hi - lo) * ((float)rand() / (float)RAND_MAX);
  float* d;
  CK(hipMalloc(&d, n * sizeof(float)));
  CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
  return d;
}
static float* dev_rand_f16pairs(size_t n) {
  std::vector<uint16_t> h(2 * n);
  for (size_t i = 0; i < 2 * n; ++i) h[i] = (uint16_t)(((rand() & 1) << 15) | ((9 + rand() % 5) << 10) | (rand() & 0x3FF));
  float* d;
  CK(hipMalloc(&d, n * sizeof(float)));
  CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
  return d;
}

int main(int argc, char** argv) {
  int rows = 2562, D = 256, H = 4, S = 3, reps = 2000;
  if (argc >= 6) { rows = atoi(argv[1]); D = atoi(argv[2]); H = atoi(argv[3]); S = atoi(argv[4]); reps = atoi(argv[5]); }
  const int DH = D / H, tiles = (rows + 31) / 32;
  hipStream_t s; CK(hipStreamCreate(&s));
  gc::GemmArgs g{};
  g.lda = D; g.a_f32 = 1; g.wt = dev_rand_f16pairs((size_t)D * D); g.ldw = D; g.rows = rows; g.n = D; g.k_slice = D;
  const size_t slots = (size_t)tiles * S * H;
  g.att_po = dev_rand(slots * 32 * DH); g.att_pml = dev_rand(slots * 64, 0.5f, 2.0f);
  g.att_S = S; g.att_B = 1; g.att_H = H; g.att_DH = DH;
  gc::RowFuse f{};
  f.x = dev_rand((size_t)rows * D); f.bias = dev_rand(D); f.cond = dev_rand(2 * D); f.cond_stride = 2 * D; f.B = 1;
  f.h = dev_rand((size_t)rows * D); f.round16 = 0;
  const bool half = tiles <= 128;
  const int wgs = half ? (rows + 15) / 16 : tiles, waves = (D / (32 * (D > 256 ? 2 : 1)));
  unsigned long long* st; CK(hipMalloc(&st, (size_t)wgs * waves * 8 * sizeof(unsigned long long)));
  CK(hipMemset(st, 0, (size_t)wgs * waves * 8 * sizeof(unsigned long long)));
  CK(gc::set_gemm_rowop_stamp_buffer(st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) CK(gc::launch_gemm_rowop(s, gc::KC_GEMM_OUT, g, f));
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < reps; ++i) CK(gc::launch_gemm_rowop(s, gc::KC_GEMM_OUT, g, f));
  CK(hipEventRecord(e1, s));
  CK(hipStreamSynchronize(s));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("rows %d D %d heads %d splits %d: %.2f us per launch (back to back), %d workgroups x %d waves\n", rows, D, H, S,
         1e3 * ms / reps, wgs, waves);
  std::vector<unsigned long long> h((size_t)wgs * waves * 8);
  CK(hipMemcpy(h.data(), st, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  std::vector<size_t> live;
  for (size_t w = 0; w < (size_t)wgs * waves; ++w) if (h[w * 8]) live.push_back(w);
  const char* names[6] = {"W ring issue + partial loads + merge + split + stage", "barrier (A tile complete)", "products (whole K)",
                          "barrier + y tile to LDS + barrier", "row pass: x loads, LN, cond, x / h stores issued", "store drain"};
  for (int p = 0; p < 6; ++p) {
    std::vector<unsigned long long> d;
    for (size_t w : live) d.push_back(h[w * 8 + p + 1] - h[w * 8 + p]);
    std::sort(d.begin(), d.end());
    printf("%-56s median %7llu  p10 %7llu  p90 %7llu  max %7llu\n", names[p], d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10], d.back());
  }
  std::vector<unsigned long long> life;
  double clk = 0;
  for (size_t w : live) { life.push_back(h[w * 8 + 6] - h[w * 8]); clk += (double)(h[w * 8 + 6] - h[w * 8]) / (double)std::max<unsigned long long>(h[w * 8 + 7], 1) * 100.0; }
  std::sort(life.begin(), life.end());
  printf("wave lifetime median %llu max %llu cycles; shader clock seen by the waves %.0f MHz; MFMA floor per wave %d cycles\n",
         life[life.size() / 2], life.back(), clk / live.size(), (D / 16) * 3 * (D > 256 ? 2 : 1) * 32);
  return 0;
}
