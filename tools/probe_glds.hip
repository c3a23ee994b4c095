// Probe: where does global_load_lds_dwordx4 put each lane's 16 bytes?  (hipcc --offload-arch=gfx950 tools/probe_glds.hip -o tools/probe_glds)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned* src, unsigned* out) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = 0xdeadbeefu;
  __syncthreads();
  const char* g = reinterpret_cast<const char*>(src) + wave * 1024 + lane * 16;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)(lds + wave * 20480), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int w = 0; w < 4; ++w) for (int i = threadIdx.x; i < 256; i += blockDim.x) out[w * 256 + i] = reinterpret_cast<unsigned*>(lds + w * 20480)[i];
}
int main() {
  std::vector<unsigned> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = i;
  unsigned *s, *o;
  hipMalloc(&s, 16384); hipMalloc(&o, 16384);
  hipMemcpy(s, h.data(), 16384, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 65536, 0, s, o);
  hipMemcpy(h.data(), o, 16384, hipMemcpyDeviceToHost);
  for (int w = 0; w < 4; ++w) {
    printf("wave %d -> lds + %d:", w, w * 20480);
    for (int i = 0; i < 256; i += 37) printf(" [%d]=%x", i, h[w * 256 + i]);
    printf("\n");
  }
  return 0;
}
