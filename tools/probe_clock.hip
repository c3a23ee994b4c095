// What does s_memtime count, and how fast is the shader clock under load?  (hipcc --offload-arch=gfx950 -O3)
// Every wave brackets its work with s_memtime and s_memrealtime (constant 100 MHz); the ratio is the s_memtime
// frequency.  Three loads: one spinning wave, all CUs spinning, all CUs issuing back-to-back fp16 MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void probe(unsigned long long* out, int mode, int iters, float* sink) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f); b[i] = (_Float16)(0.5f); }
  float x = threadIdx.x;
  if (mode == 2) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
    }
  } else {
    for (int it = 0; it < iters; ++it) x = x * 1.0000001f + 0.5f;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {
    unsigned long long* o = out + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2;
    o[0] = t1 - t0; o[1] = r1 - r0;
  }
  float s = x;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
  if (s == 12345.678f) sink[0] = s;
}
int main() {
  unsigned long long* d; float* sink;
  hipMalloc(&d, 1024 * 16 * 2 * sizeof(unsigned long long)); hipMalloc(&sink, 4);
  struct { const char* name; int mode, blocks, threads, iters; } runs[] = {
      {"one wave, scalar FMA chain", 0, 1, 64, 2000000},
      {"256 x 4 waves, scalar FMA chain", 0, 256, 256, 2000000},
      {"256 x 4 waves, back-to-back 32x32x16 fp16 MFMA", 2, 256, 256, 200000},
      {"512 x 4 waves (2 per SIMD), back-to-back MFMA", 2, 512, 256, 200000},
      {"one wave, back-to-back MFMA", 2, 1, 64, 200000}};
  for (auto& r : runs) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<<<r.blocks, r.threads>>>(d, r.mode, 1000, sink);   // warm
    hipEventRecord(e0);
    probe<<<r.blocks, r.threads>>>(d, r.mode, r.iters, sink);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int waves = r.blocks * (r.threads / 64);
    std::vector<unsigned long long> h(waves * 2);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::vector<double> ratio;
    for (int w = 0; w < waves; ++w) ratio.push_back((double)h[2 * w] / (double)h[2 * w + 1]);
    std::sort(ratio.begin(), ratio.end());
    double mf = 0;
    if (r.mode == 2) mf = (double)waves * r.iters * 4 * 32768.0 / (ms * 1e-3) / 1e12;
    printf("%-52s %8.2f ms  s_memtime ticks %llu  s_memrealtime ticks %llu  ratio median %.3f (min %.3f max %.3f) -> s_memtime at %.0f MHz",
           r.name, ms, h[0], h[1], ratio[ratio.size() / 2], ratio.front(), ratio.back(), ratio[ratio.size() / 2] * 100.0);
    if (mf > 0) printf("  | %.0f TFLOP/s dense fp16; cycles per MFMA per wave %.1f s_memtime ticks", mf, (double)h[0] / (r.iters * 4.0));
    printf("\n");
  }
  return 0;
}
