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

// barrier-free variant: every wave streams its own A / W fragments straight from global (L2)
// into registers, DEPTH k-tiles ahead; no LDS.
template <int DEPTH>
__global__ __launch_bounds__(256) void k_direct(float* out, const float* src, int iters, int src_rows) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  f32x16 acc;
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  size_t base = ((size_t)blockIdx.x * 37) % (size_t)(src_rows - 200);
  const float* ap = src + (base + r) * 256 + hh * 16;                 // A rows shared by the 4 waves
  const float* wp = src + (base + 40 + wave * 32 + r) * 256 + hh * 16; // W rows private per wave
  f32x4 fa[DEPTH][4], fw[DEPTH][4];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int q = 0; q < 4; ++q) { fa[d][q] = *(const f32x4*)(ap + d * 32 + 4 * q); fw[d][q] = *(const f32x4*)(wp + d * 32 + 4 * q); }
  for (int it = 0; it < iters; it += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      f32x4 ca[4], cw[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { ca[q] = fa[d][q]; cw[q] = fw[d][q]; }
      const int ko = ((it + d + DEPTH) & 7) * 32;
#pragma unroll
      for (int q = 0; q < 4; ++q) { fa[d][q] = *(const f32x4*)(ap + ko + 4 * q); fw[d][q] = *(const f32x4*)(wp + ko + 4 * q); }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[q][e], cw[q][e], acc, 0, 0, 0);
      if (((it + d) & 7) == 7) {
        const size_t orow = ((size_t)blockIdx.x * 131 + ((it + d) >> 3) * 977) % 2500;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          out[(orow + (q & 3) + 8 * (q >> 2) + 4 * hh) * 2048 + (blockIdx.x % 16) * 128 + wave * 32 + r] = acc[q] + 1.0f;
          acc[q] = 0.f;
        }
        base = (base + 977) % (size_t)(src_rows - 200);
        ap = src + (base + r) * 256 + hh * 16;
        wp = src + (base + 40 + wave * 32 + r) * 256 + hh * 16;
      }
    }
  }
  float s = 0;
  for (int q = 0; q < 16; ++q) s += acc[q];
  if (s == 12345.678f) out[tid] = s;
}

// LDS-DMA variant: global_load_lds_dwordx4 into a 3-deep ring (tile c+1, c+2 in flight during the
// MFMAs of tile c), XOR-swizzled 128-byte rows, counted vmcnt + raw s_barrier.
template <bool E, int RING>
__global__ __launch_bounds__(256) void k_dma(float* out, const float* src, int iters, int src_rows) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  constexpr int TILE = 160 * 32;                       // floats per ring slot (A 32 rows + W 128 rows)
  extern __shared__ __attribute__((aligned(1024))) float ring[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  f32x16 acc;
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  size_t base = ((size_t)blockIdx.x * 37) % (size_t)(src_rows - 160);
  // DMA role: wave w fills rows [40w, 40w+40) of the slot = 5 instructions of 8 rows each
  const int drow = lane >> 3, dchunk = lane & 7;
  // read role: logical chunk c of row R lives at physical chunk c ^ ((R >> 1) & 7)
  int a_off[4], b_off[4];
  for (int q = 0; q < 4; ++q) {
    const int ra = r, rb = 32 + wave * 32 + r;
    a_off[q] = ra * 32 + (((hh * 4 + q) ^ ((ra >> 1) & 7)) * 4);
    b_off[q] = rb * 32 + (((hh * 4 + q) ^ ((rb >> 1) & 7)) * 4);
  }
  auto issue = [&](int it) {
    const int slot = it % RING, ko = (it & 7) * 32;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int row = wave * 40 + i * 8 + drow;
      const int c = dchunk ^ ((row >> 1) & 7);
      const float* g = src + (base + row) * 256 + ko + c * 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(ring + slot * TILE + (wave * 40 + i * 8) * 32),
                                       16, 0, 0);
    }
  };
  for (int d = 0; d < RING - 1; ++d) issue(d);
  for (int it = 0; it < iters; ++it) {
    // tiles it+1 .. it+RING-2 may stay in flight (5 DMA instructions each)
    if (it + RING - 2 < iters) {
      if (RING == 3) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      if (RING == 4) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      if (RING == 5) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
      if (RING == 6) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (it + RING - 1 < iters) issue(it + RING - 1);
    const float* sl = ring + (it % RING) * TILE;
    f32x4 a4[4], b4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { a4[q] = *(const f32x4*)(sl + a_off[q]); b4[q] = *(const f32x4*)(sl + b_off[q]); }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q][e], b4[q][e], acc, 0, 0, 0);
    if (E && (it & 7) == 7) {
      const size_t orow = ((size_t)blockIdx.x * 131 + (it >> 3) * 977) % 2500;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        out[(orow + (q & 3) + 8 * (q >> 2) + 4 * hh) * 2048 + (blockIdx.x % 16) * 128 + wave * 32 + r] = acc[q] + 1.0f;
        acc[q] = 0.f;
      }
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
  float* out; CK(hipMalloc(&out, (size_t)2600 * 2048 * 4 + (1 << 24)));
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
#define DMA(R) { hipFuncSetAttribute((const void*)k_dma<true, R>, hipFuncAttributeMaxDynamicSharedMemorySize, R * 160 * 32 * 4); \
    rep("lds-dma ring" #R " +epilogue", run([&] { hipLaunchKernelGGL((k_dma<true, R>), dim3(nb), dim3(256), R * 160 * 32 * 4, 0, out, src, iters, 4700); }), 1); }
    DMA(3) DMA(4) DMA(5) DMA(6)
    rep("direct depth2 (+epilogue)", run([&] { hipLaunchKernelGGL((k_direct<2>), dim3(nb), dim3(256), 0, 0, out, src, iters, 4700); }), 1);
    rep("direct depth4 (+epilogue)", run([&] { hipLaunchKernelGGL((k_direct<4>), dim3(nb), dim3(256), 0, 0, out, src, iters, 4700); }), 1);
    rep("full: dswrite only", run([&] { hipLaunchKernelGGL((k_full<false, true, false>), dim3(nb), dim3(256), 0, 0, out, src, iters, 4700); }), 1);
    rep("full: epilogue only", run([&] { hipLaunchKernelGGL((k_full<false, false, true>), dim3(nb), dim3(256), 0, 0, out, src, iters, 4700); }), 1);
    rep("tile 2acc lds+barrier", run([&] { hipLaunchKernelGGL((k_tile<2, true, true>), dim3(nb), dim3(256), 0, 0, out, iters); }), 2);
  }
  return 0;
}
