// Isotropic spherical white noise on the device.
//
// Reference: gencast/samplers_utils.py:250-346 (`sample` / `spherical_white_noise_like`): Gaussian
// coefficients c_lm ~ N(0,1) for |m| <= l < L = n_lon/2, scaled by sqrt(4 pi p_l / (2l+1)), inverse
// real-spherical-harmonic transform onto the equiangular lat/lon grid.  The transform is separable:
//   Legendre step  F_cos[m][lat][n] = sum_l  Pn[m][lat][l] * c_cos[m][l][n]     (same for sin)
//   Fourier step   x[lat][lon][n]   = sum_m  C[lon][m] F_cos[m][lat][n] + S[lon][m] F_sin[m][lat][n]
// with the tables Pn (normalised associated Legendre functions times the per-l factor; zero for l < m),
// C / S (cos / sin of m * lon, times sqrt(2) for m > 0) built once on the host
// (gencast-flax-nnx_amd/noise.py).  n = batch * channel: every column is an independent field.
// Both steps are tiny (0.3 GFLOP per 82-channel 2.5-degree field) and run on the vector ALUs.
//
// Random numbers: Philox4x32-10 (Salmon et al., SC'11), counter = (group, stream), key = seed; four
// 32-bit words -> two Box-Muller pairs.  Counter-based, so a field depends only on (seed, stream),
// never on launch geometry; oracle/noise_oracle.py restates the generator bit for bit.
#include "gc_kernels.h"

#include <math.h>

namespace gc {

__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
  const unsigned long long p0 = 0xD2511F53ull * c[0];
  const unsigned long long p1 = 0xCD9E8D57ull * c[2];
  const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0;
  const unsigned n1 = (unsigned)p1;
  const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1;
  const unsigned n3 = (unsigned)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

// out[4 g .. 4 g + 3] = N(0,1) from Philox group g of `stream`
__global__ __launch_bounds__(256) void gc_noise_normals_kernel(float* __restrict__ out, size_t count,
                                                                unsigned k0, unsigned k1, unsigned s0,
                                                                unsigned s1) {
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (4 * g >= count) return;
  unsigned c[4] = {(unsigned)g, (unsigned)(g >> 32), s0, s1};
  philox4x32_10(c, k0, k1);
  float z[4];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const float u1 = ((float)c[2 * p] + 0.5f) * 2.3283064365386963e-10f;       // (0, 1]
    const float u2 = ((float)c[2 * p + 1] + 0.5f) * 2.3283064365386963e-10f;
    const float rad = sqrtf(-2.0f * logf(fminf(u1, 1.0f)));
    float sn, cs;
    sincosf(6.283185307179586f * u2, &sn, &cs);
    z[2 * p] = rad * cs;
    z[2 * p + 1] = rad * sn;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (4 * g + e < count) out[4 * g + e] = z[e];
}

// Legendre step.  grid = (L, 2 * latitude blocks, ceil(N / 64)); 4 waves: wave w owns latitudes lat0 + w, lat0 + w + 4, ...
// of its block of kNoiseLatBlock latitudes; lane = column n.  Any n_lat (0.25 degree: 721 = 4 blocks).
constexpr int kNoiseMaxLatPerWave = 48;
constexpr int kNoiseLatBlock = 4 * kNoiseMaxLatPerWave;   // 192 latitudes per workgroup
__global__ __launch_bounds__(256) void gc_noise_legendre_kernel(const float* __restrict__ leg,   // [L][n_lat][L]
                                                                 const float* __restrict__ coef,  // [2][L][L][N]
                                                                 int L, int n_lat, int N,
                                                                 float* __restrict__ f) {         // [2][L][n_lat][N]
  const int m = blockIdx.x, part = blockIdx.y & 1, lat0 = (blockIdx.y >> 1) * kNoiseLatBlock;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = blockIdx.z * 64 + lane;
  float acc[kNoiseMaxLatPerWave];
#pragma unroll
  for (int i = 0; i < kNoiseMaxLatPerWave; ++i) acc[i] = 0.f;
  const float* cm = coef + ((size_t)part * L + m) * L * N;
  const float* lm = leg + (size_t)m * n_lat * L;
  for (int l = m; l < L; ++l) {
    const float cv = (n < N) ? cm[(size_t)l * N + n] : 0.f;
#pragma unroll
    for (int i = 0; i < kNoiseMaxLatPerWave; ++i) {
      const int lat = lat0 + wave + 4 * i;
      if (lat < n_lat) acc[i] += lm[(size_t)lat * L + l] * cv;     // wave-uniform table value
    }
  }
  if (n >= N) return;
  float* fm = f + ((size_t)part * L + m) * n_lat * N;
#pragma unroll
  for (int i = 0; i < kNoiseMaxLatPerWave; ++i) {
    const int lat = lat0 + wave + 4 * i;
    if (lat < n_lat) fm[(size_t)lat * N + n] = acc[i];
  }
}

// Fourier step + the consumer's update:  out[node][n] = (base ? base[node][n] : 0) + scale * x[node][n].
// grid = (n_lat, ceil(n_lon / 32), ceil(N / 64)); wave w owns 8 longitudes, lane = column n of the workgroup's 64-column
// group.  The latitude's coefficients F[part][m][n] go through LDS in chunks of kNoiseMChunk wavenumbers x 64 columns
// (64 KB), whatever L and N are -- round 4 staged all of [2][L][N] at once and refused 2 L N floats > 160 KB (1 degree with
// batch 2, any finer grid).  m is summed in ascending order across the chunks: the same additions as the one-chunk form.
constexpr int kNoiseMChunk = 128;
__global__ __launch_bounds__(256) void gc_noise_fourier_kernel(const float* __restrict__ f,      // [2][L][n_lat][N]
                                                                const float* __restrict__ ctab,   // [n_lon][L]
                                                                const float* __restrict__ stab,   // [n_lon][L]
                                                                int L, int n_lat, int n_lon, int N,
                                                                const float* __restrict__ base, float scale,
                                                                float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float fs_lds[2 * kNoiseMChunk * 64];   // [2][chunk][64]
  const int lat = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n0 = blockIdx.z * 64, n = n0 + lane;
  const int lon0 = blockIdx.y * 32 + wave * 8;
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  for (int m0 = 0; m0 < L; m0 += kNoiseMChunk) {
    const int mc = (L - m0) < kNoiseMChunk ? (L - m0) : kNoiseMChunk;
    if (m0) __syncthreads();                          // every wave is done with the previous chunk
    for (int i = threadIdx.x; i < 2 * mc * 64; i += 256) {
      const int part = i / (mc * 64), r = i - part * mc * 64, m = r >> 6, c = r & 63;
      fs_lds[(part * kNoiseMChunk + m) * 64 + c] =
          (n0 + c < N) ? f[(((size_t)part * L + m0 + m) * n_lat + lat) * N + n0 + c] : 0.f;
    }
    __syncthreads();
    for (int m = 0; m < mc; ++m) {
      const float a = fs_lds[m * 64 + lane];
      const float b = fs_lds[(kNoiseMChunk + m) * 64 + lane];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int lon = lon0 + k < n_lon ? lon0 + k : n_lon - 1;
        acc[k] += ctab[(size_t)lon * L + m0 + m] * a + stab[(size_t)lon * L + m0 + m] * b;
      }
    }
  }
  if (n < N) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int lon = lon0 + k;
      if (lon < n_lon) {
        const size_t i = ((size_t)lat * n_lon + lon) * N + n;
        out[i] = (base ? base[i] : 0.f) + scale * acc[k];
      }
    }
  }
}

hipError_t launch_noise_normals(hipStream_t s, float* out, size_t count, unsigned long long key,
                                unsigned long long stream) {
  const size_t groups = (count + 3) / 4;
  if (groups == 0) return hipSuccess;
  hipLaunchKernelGGL(gc_noise_normals_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, out, count,
                     (unsigned)key, (unsigned)(key >> 32), (unsigned)stream, (unsigned)(stream >> 32));
  return hipGetLastError();
}

hipError_t launch_noise_synthesis(hipStream_t s, const float* leg, const float* ctab, const float* stab,
                                  const float* coef, float* f, int L, int n_lat, int n_lon, int N,
                                  const float* base, float scale, float* out) {
  if (L < 1 || n_lat < 1 || n_lon < 1 || N < 1) return hipErrorInvalidValue;
  const unsigned lat_blocks = (unsigned)((n_lat + kNoiseLatBlock - 1) / kNoiseLatBlock), col_groups = (unsigned)((N + 63) / 64);
  if (2 * lat_blocks > 65535u || col_groups > 65535u || (unsigned)((n_lon + 31) / 32) > 65535u) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gc_noise_legendre_kernel, dim3(L, 2 * lat_blocks, col_groups), dim3(256), 0, s, leg, coef, L, n_lat, N, f);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(gc_noise_fourier_kernel, dim3(n_lat, (n_lon + 31) / 32, col_groups), dim3(256), 0, s, f, ctab, stab, L,
                     n_lat, n_lon, N, base, scale, out);
  return hipGetLastError();
}

}  // namespace gc
