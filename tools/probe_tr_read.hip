#include <hip/hip_runtime.h>
typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__global__ void k(const _Float16* in, _Float16* out) {
  __shared__ _Float16 lds[64 * 64];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const _Float16* addr = lds + (4 * g + q) * 64 + 4 * p;
  h4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)addr);
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (_Float16)v[e];
}
int main() {
  _Float16 h[4096], o[256];
  for (int i = 0; i < 4096; ++i) h[i] = (_Float16)(i / 64 * 100 + i % 64);   // value = row*100 + col
  _Float16 *di, *d_o;
  hipMalloc(&di, sizeof(h)); hipMalloc(&d_o, sizeof(o));
  hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, d_o);
  hipMemcpy(o, d_o, sizeof(o), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 1) { printf("lane %2d:", l); for (int e = 0; e < 4; ++e) printf(" %6.0f", (float)o[l * 4 + e]); printf("\n"); }
  return 0;
}
