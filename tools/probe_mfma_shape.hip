// 32x32x16 against 16x16x32 fp16 MFMA under load (the shape as a lever where the chip holds its
// clock down).  Each wave multiplies a 64 x 64 x 32 update per step from LDS-resident operands (8 ds_read_b128 per step,
// the same bytes for both shapes; 64 accumulator registers both): 8 x v_mfma_f32_32x32x16_f16 or 16 x v_mfma_f32_16x16x32_f16.
// Reported per run: wall time, TFLOP/s, cycles per step (s_memtime) and the shader clock (s_memtime / s_memrealtime) seen by
// the waves -- random operands and zeros, one and two waves per SIMD.   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_shape.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE /* 32 or 16 */, int STEPS_IN_LDS>
__global__ __launch_bounds__(256) void probe(const _Float16* __restrict__ src, unsigned long long* stamps, float* sink, int iters) {
  // per wave: STEPS_IN_LDS steps x (A 64x32 + B 64x32 halfs) = STEPS_IN_LDS x 8 KB
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  _Float16* mine = lds + (size_t)wave * STEPS_IN_LDS * 4096;
  for (int i = lane * 8; i < STEPS_IN_LDS * 4096; i += 64 * 8)
    *reinterpret_cast<f16x8*>(mine + i) = *reinterpret_cast<const f16x8*>(src + ((size_t)(blockIdx.x * 4 + wave) * STEPS_IN_LDS * 4096 + i) % (1 << 24));
  __syncthreads();
  f32x16 acc32[4];
  f32x4 acc16[16];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc32[j][i] = 0.f;
  for (int j = 0; j < 16; ++j) for (int i = 0; i < 4; ++i) acc16[j][i] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  // fragment images: 16-byte pieces in lane order (conflict-free ds_read_b128); 4 pieces of A, 4 of B per step.  The pieces
  // of step s + 1 are requested before the MFMAs of step s (two register sets), so the matrix pipe does not wait for LDS.
  f16x8 fa[2][4], fb[2][4];
  auto fetch = [&](int set, int s, int it) __attribute__((always_inline)) {
    const _Float16* p = mine + ((s + it) & (STEPS_IN_LDS - 1)) * 4096 + lane * 8;   // varies with `it`: the reads stay in the loop
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      fa[set][q] = *reinterpret_cast<const f16x8*>(p + q * 512);
      fb[set][q] = *reinterpret_cast<const f16x8*>(p + 2048 + q * 512);
    }
  };
  fetch(0, 0, 0);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < STEPS_IN_LDS; ++s) {
      const int cur = s & 1;
      fetch(cur ^ 1, s + 1, it);
      const f16x8 (&a)[4] = fa[cur];
      const f16x8 (&b)[4] = fb[cur];
      if constexpr (SHAPE == 32) {
        // a[2 * ks + mt], b[2 * ks + nt]: two k16 steps of a 2 x 2 arrangement of 32 x 32 tiles
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
              acc32[mt * 2 + nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 * ks + mt], b[2 * ks + nt], acc32[mt * 2 + nt], 0, 0, 0);
      } else {
        // a[mt], b[nt]: one k32 step of a 4 x 4 arrangement of 16 x 16 tiles
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            acc16[mt * 4 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mt], b[nt], acc16[mt * 4 + nt], 0, 0, 0);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) {
    unsigned long long* o = stamps + ((size_t)blockIdx.x * 4 + wave) * 2;
    o[0] = t1 - t0;
    o[1] = r1 - r0;
  }
  float sum = 0.f;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) sum += acc32[j][i];
  for (int j = 0; j < 16; ++j) for (int i = 0; i < 4; ++i) sum += acc16[j][i];
  if (sum == 12345.678f) sink[0] = sum;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int SHAPE>
void run(const char* what, const _Float16* src, unsigned long long* stamps, float* sink, int blocks, int iters) {
  constexpr int STEPS = 4;
  const size_t ldsb = (size_t)4 * STEPS * 4096 * sizeof(_Float16);     // 128 KB at one workgroup per CU, 2 x 64 KB at two
  CK(hipFuncSetAttribute((const void*)probe<SHAPE, STEPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
  // keep the chip under this load for ~2 s before the measured launch (the clock follows the load with a delay)
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 12; ++w) hipLaunchKernelGGL((probe<SHAPE, STEPS>), dim3(blocks), dim3(256), ldsb, 0, src, stamps, sink, iters);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((probe<SHAPE, STEPS>), dim3(blocks), dim3(256), ldsb, 0, src, stamps, sink, iters);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const int waves = blocks * 4;
  std::vector<unsigned long long> h((size_t)waves * 2);
  CK(hipMemcpy(h.data(), stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  std::vector<double> clk, cyc;
  for (int w = 0; w < waves; ++w) {
    clk.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 100.0);
    cyc.push_back((double)h[2 * w] / ((double)iters * STEPS));
  }
  std::sort(clk.begin(), clk.end());
  std::sort(cyc.begin(), cyc.end());
  const double flop = (double)waves * iters * STEPS * 2.0 * 64 * 64 * 32;
  printf("%-58s %7.2f ms  %7.1f TFLOP/s  cycles per 64x64x32 step median %6.1f  clock median %5.0f MHz (min %4.0f max %4.0f)\n",
         what, ms, flop / (ms * 1e-3) / 1e12, cyc[cyc.size() / 2], clk[clk.size() / 2], clk.front(), clk.back());
}

int main() {
  _Float16* src; unsigned long long* stamps; float* sink;
  const size_t n = (size_t)1 << 24;
  CK(hipMalloc(&src, n * sizeof(_Float16) + (1 << 20))); CK(hipMalloc(&stamps, 4096 * 2 * sizeof(unsigned long long))); CK(hipMalloc(&sink, 4));
  std::vector<_Float16> h(n + (1 << 19));
  for (int pass = 0; pass < 2; ++pass) {
    srand(1);
    for (auto& v : h) v = pass == 0 ? (_Float16)((rand() % 2001 - 1000) * 1e-3f) : (_Float16)0.f;
    CK(hipMemcpy(src, h.data(), h.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    const char* d = pass == 0 ? "random operands" : "zero operands";
    char buf[128];
    snprintf(buf, sizeof buf, "32x32x16, %s, 1 wave per SIMD (256 workgroups)", d);
    run<32>(buf, src, stamps, sink, 256, 150000);
    snprintf(buf, sizeof buf, "16x16x32, %s, 1 wave per SIMD (256 workgroups)", d);
    run<16>(buf, src, stamps, sink, 256, 150000);
    snprintf(buf, sizeof buf, "32x32x16, %s, 1 wave per SIMD, again", d);
    run<32>(buf, src, stamps, sink, 256, 150000);
    snprintf(buf, sizeof buf, "16x16x32, %s, 1 wave per SIMD, again", d);
    run<16>(buf, src, stamps, sink, 256, 150000);
  }
  return 0;
}
