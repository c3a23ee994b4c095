// Probe (round 5): is v_mfma_f32_32x32x16_f16 accumulation independent of how the instruction stream interleaves two
// accumulator sets?  One wave runs the product pattern of ws_quad (acc2 += wl*ah; acc += wh*ah; acc2 += wh*al) for two
// "row tiles" mt = 0, 1 that are given IDENTICAL operands, in the kernel's order (all of mt 0, then all of mt 1, per k16
// step), and compares the two accumulator sets bit for bit.  Also: the same chain for mt 1 issued with the operands of a step
// loaded through LDS right behind mt 0's MFMAs (the kernel's register reuse).
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_order.hip -o tools/probe_mfma_order && tools/probe_mfma_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma16(f32x4 a, f32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__global__ void probe(const f32x4* wh, const f32x4* wl, const f32x4* ah, const f32x4* al, int steps, float* out, int via_lds) {
  __shared__ f32x4 lds[2][2][64];
  const int lane = threadIdx.x;
  f32x16 acc[2], acc2[2];
  for (int m = 0; m < 2; ++m) for (int g = 0; g < 16; ++g) { acc[m][g] = 0.f; acc2[m][g] = 0.f; }
  for (int s = 0; s < steps; ++s) {
    const f32x4 h = wh[s * 64 + lane], l = wl[s * 64 + lane];
    f32x4 bh = ah[s * 64 + lane], bl = al[s * 64 + lane];
    if (via_lds) { lds[0][0][lane] = bh; lds[0][1][lane] = bl; lds[1][0][lane] = bh; lds[1][1][lane] = bl; __syncthreads(); }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      if (via_lds) { bh = lds[m][0][lane]; bl = lds[m][1][lane]; }
      acc2[m] = mfma16(l, bh, acc2[m]);
      acc[m] = mfma16(h, bh, acc[m]);
      acc2[m] = mfma16(h, bl, acc2[m]);
    }
    if (via_lds) __syncthreads();
  }
  for (int m = 0; m < 2; ++m)
    for (int g = 0; g < 16; ++g) {
      out[((m * 2 + 0) * 16 + g) * 64 + lane] = acc[m][g];
      out[((m * 2 + 1) * 16 + g) * 64 + lane] = acc2[m][g];
    }
}
int main() {
  const int steps = 32;
  std::vector<_Float16> h(4 * steps * 64 * 8);
  srand(1);
  for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.f);
  _Float16* d; float* o;
  hipMalloc(&d, h.size() * 2); hipMalloc(&o, 4 * 16 * 64 * 4);
  hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  const size_t n = (size_t)steps * 64;
  for (int via = 0; via < 2; ++via) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, (const f32x4*)d, (const f32x4*)d + n, (const f32x4*)d + 2 * n, (const f32x4*)d + 3 * n, steps, o, via);
    std::vector<float> r(4 * 16 * 64);
    hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
    int da = 0, d2 = 0;
    for (int i = 0; i < 16 * 64; ++i) {
      da += memcmp(&r[i], &r[2 * 16 * 64 + i], 4) != 0;
      d2 += memcmp(&r[16 * 64 + i], &r[3 * 16 * 64 + i], 4) != 0;
    }
    printf("via_lds %d: acc differs in %d of 1024, acc2 differs in %d of 1024\n", via, da, d2);
  }
  return 0;
}
