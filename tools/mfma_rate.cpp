// Raw issue-rate probes for v_mfma_f32_32x32x2_f32 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void k_chain(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int q = 0; q < 16; ++q) s += acc[i][q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// same chain but with LDS fragment reads + a barrier every 16 MFMAs (the GEMM k-tile skeleton)
template <int NACC, bool BARRIER, bool LDSREAD>
__global__ __launch_bounds__(256) void k_tile(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[2][160][36];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  for (int i = tid; i < 2 * 160 * 36; i += 256) (&lds[0][0][0])[i] = 1e-3f * (i % 97);
  __syncthreads();
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 a4[NACC][4], b4[4];
  for (int q = 0; q < 4; ++q) { b4[q] = *(f32x4*)&lds[0][32 + wave * 32 + r][hh * 16 + 4 * q]; for (int i = 0; i < NACC; ++i) a4[i][q] = *(f32x4*)&lds[0][(i & 0) * 32 + r][hh * 16 + 4 * q]; }
  for (int it = 0; it < iters; ++it) {
    const int buf = it & 1;
    if (LDSREAD) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        b4[q] = *(f32x4*)&lds[buf][32 + wave * 32 + r][hh * 16 + 4 * q];
#pragma unroll
        for (int i = 0; i < NACC; ++i) a4[i][q] = *(f32x4*)&lds[buf][r][hh * 16 + 4 * q];
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i][q][e], b4[q][e], acc[i], 0, 0, 0);
    if (BARRIER) __syncthreads();
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int q = 0; q < 16; ++q) s += acc[i][q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// full k-tile skeleton: optional global loads (G), LDS staging writes (W), epilogue every 8 tiles (E)
template <bool G, bool W, bool E>
__global__ __launch_bounds__(256) void k_full(float* out, const float* src, int iters, int src_rows) {
  __shared__ __attribute__((aligned(16))) float lds[2][160][36];
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int lrow = tid >> 3, lc4 = tid & 7;
  for (int i = tid; i < 2 * 160 * 36; i += 256) (&lds[0][0][0])[i] = 1e-3f * (i % 97);
  __syncthreads();
  f32x16 acc;
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  f32x4 rg[5];
  for (int i = 0; i < 5; ++i) rg[i] = f32x4{0, 0, 0, 0};
  size_t base = ((size_t)blockIdx.x * 37) % (size_t)(src_rows - 160);
  for (int it = 0; it < iters; ++it) {
    const int buf = it & 1;
    if (G) {
      const int ko = (it & 7) * 32;
#pragma unroll
      for (int i = 0; i < 5; ++i) rg[i] = *(const f32x4*)(src + (base + lrow + 32 * i) * 256 + ko + lc4 * 4);
    }
    f32x4 a4[4], b4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      b4[q] = *(f32x4*)&lds[buf][32 + wave * 32 + r][hh * 16 + 4 * q];
      a4[q] = *(f32x4*)&lds[buf][r][hh * 16 + 4 * q];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q][e], b4[q][e], acc, 0, 0, 0);
    if (W) {
#pragma unroll
      for (int i = 0; i < 5; ++i) *(f32x4*)&lds[buf ^ 1][lrow + 32 * i][lc4 * 4] = rg[i];
    }
    __syncthreads();
    if (E && (it & 7) == 7) {
      const size_t orow = ((size_t)blockIdx.x * 131 + (it >> 3) * 977) % 2500;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        out[(orow + (q & 3) + 8 * (q >> 2) + 4 * hh) * 2048 + (blockIdx.x % 16) * 128 + wave * 32 + r] = acc[q] + 1.0f;
        acc[q] = 0.f;
      }
      base = (base + 977) % (size_t)(src_rows - 160);
    }
  }
  float s = 0;
  for (int q = 0; q < 16; ++q) s += acc[q];
  if (s == 12345.678f) out[tid] = s;
}

template <typename F> float run(F f) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms;
}

int main() {
  float* out; CK(hipMalloc(&out, (size_t)2600 * 2048 * 4));
  float* src; CK(hipMalloc(&src, (size_t)4700 * 256 * 4)); CK(hipMemset(src, 0, (size_t)4700 * 256 * 4));
  const int iters = 2000;
  for (int blocks_per_cu : {1, 2, 3}) {
    const int nb = 256 * blocks_per_cu;
    auto rep = [&](const char* name, float ms, int nacc) {
      double mf = (double)iters * 16 * nacc;            // MFMAs per wave
      double cyc_per = ms * 1e-3 * 2.4e9 / (mf * blocks_per_cu);   // per MFMA per SIMD if pipe-bound
      double tf = (double)nb * 4 * mf * 4096 / (ms * 1e-3) / 1e12;
      printf("blocks/CU %d  %-34s %8.3f ms  %6.1f TF/s  (%.1f cyc/MFMA/SIMD @2.4GHz)\n", blocks_per_cu, name, ms, tf, cyc_per);
    };
    rep("chain 1 acc", run([&] { hipLaunchKernelGGL(k_chain<1>, dim3(nb), dim3(256), 0, 0, out, iters, 1.f, 2.f); }), 1);
    rep("chain 2 acc", run([&] { hipLaunchKernelGGL(k_chain<2>, dim3(nb), dim3(256), 0, 0, out, iters, 1.f, 2.f); }), 2);
    rep("tile 1acc lds+barrier", run([&] { hipLaunchKernelGGL((k_tile<1, true, true>), dim3(nb), dim3(256), 0, 0, out, iters); }), 1);
    rep("tile 1acc lds only", run([&] { hipLaunchKernelGGL((k_tile<1, false, true>), dim3(nb), dim3(256), 0, 0, out, iters); }), 1);
    rep("tile 1acc barrier only", run([&] { hipLaunchKernelGGL((k_tile<1, true, false>), dim3(nb), dim3(256), 0, 0, out, iters); }), 1);
    rep("full: skeleton", run([&] { hipLaunchKernelGGL((k_full<false, false, false>), dim3(nb), dim3(256), 0, 0, out, src, iters, 4700); }), 1);
    rep("full: +gload", run([&] { hipLaunchKernelGGL((k_full<true, false, false>), dim3(nb), dim3(256), 0, 0, out, src, iters, 4700); }), 1);
    rep("full: +gload+dswrite", run([&] { hipLaunchKernelGGL((k_full<true, true, false>), dim3(nb), dim3(256), 0, 0, out, src, iters, 4700); }), 1);
    rep("full: +gload+dswrite+epilogue", run([&] { hipLaunchKernelGGL((k_full<true, true, true>), dim3(nb), dim3(256), 0, 0, out, src, iters, 4700); }), 1);
    rep("full: dswrite only", run([&] { hipLaunchKernelGGL((k_full<false, true, false>), dim3(nb), dim3(256), 0, 0, out, src, iters, 4700); }), 1);
    rep("full: epilogue only", run([&] { hipLaunchKernelGGL((k_full<false, false, true>), dim3(nb), dim3(256), 0, 0, out, src, iters, 4700); }), 1);
    rep("tile 2acc lds+barrier", run([&] { hipLaunchKernelGGL((k_tile<2, true, true>), dim3(nb), dim3(256), 0, 0, out, iters); }), 2);
  }
  return 0;
}
