"""CPU restatement of the device noise generator and the stochastic-churn sampler (TEST INFRASTRUCTURE
ONLY: only tests/ may import this).

  * `philox4x32_10`, `philox_normals`: the counter-based generator of csrc/gc_noise.hip, bit for bit
    (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; known-answer
    vectors of the Random123 distribution are checked in tests/test_noise.py) and its Box-Muller step
    in the same float32 operations.
  * `spherical_field`: gencast/samplers_utils.py:250-318 evaluated DIRECTLY,
        x(lat, lon) = sqrt(4 pi) sum_l sqrt(p_l / (2l+1)) sum_m c_lm Y_lm(lat, lon),
    with real orthonormal harmonics built from scipy.special.lpmv (Condon-Shortley phase included) --
    no shared code with the product's factorised tables (gencast-flax-nnx_amd/noise.py).
  * `dpm_solver_2s_sample_churn`: gencast/dpm_solver_plus_plus_2s.py:120-158 with the churn branch
    (:128-137) doing what `apply_stochastic_churn` does (gencast/samplers_utils.py:434-452).
Parity status: unpinned (dinosaur / jax.random absent; the reference promises the spectrum only in
expectation) -- the coefficient ordering [part][m][l][column] and the Philox streams are this build's.
"""
import numpy as np
import scipy.special

from oracle import gencast_oracle as O

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85


def philox4x32_10(counter: np.ndarray, key) -> np.ndarray:
  """counter [..., 4] uint32, key (k0, k1) -> [..., 4] uint32."""
  c = np.asarray(counter, dtype=np.uint32).copy()
  k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
  for _ in range(10):
    p0 = _M0 * c[..., 0].astype(np.uint64)
    p1 = _M1 * c[..., 2].astype(np.uint64)
    n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c[..., 1] ^ np.uint32(k0)
    n1 = p1.astype(np.uint32)
    n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c[..., 3] ^ np.uint32(k1)
    n3 = p0.astype(np.uint32)
    c = np.stack([n0, n1, n2, n3], axis=-1)
    k0 = (k0 + _W0) & 0xFFFFFFFF
    k1 = (k1 + _W1) & 0xFFFFFFFF
  return c


def philox_normals(count: int, seed: int, stream: int) -> np.ndarray:
  """`count` N(0,1) float32 values exactly as gc_noise_normals_kernel lays them out."""
  groups = (count + 3) // 4
  g = np.arange(groups, dtype=np.uint64)
  ctr = np.stack([(g & np.uint64(0xFFFFFFFF)).astype(np.uint32), (g >> np.uint64(32)).astype(np.uint32),
                  np.full(groups, stream & 0xFFFFFFFF, np.uint32), np.full(groups, (stream >> 32) & 0xFFFFFFFF, np.uint32)],
                 axis=-1)
  w = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
  scale = np.float32(2.3283064365386963e-10)
  u = (w.astype(np.float32) + np.float32(0.5)) * scale
  u1, u2 = np.minimum(u[:, 0::2], np.float32(1.0)), u[:, 1::2]
  rad = np.sqrt(np.float32(-2.0) * np.log(u1))
  ang = np.float32(6.283185307179586) * u2
  z = np.stack([rad * np.cos(ang), rad * np.sin(ang)], axis=-1)          # [groups, 2 pairs, (cos, sin)]
  return z.reshape(-1)[:count].astype(np.float32)


def real_harmonic(l: int, m: int, lat_deg: np.ndarray, lon_deg: np.ndarray) -> np.ndarray:
  """Real orthonormal Y_lm on the unit sphere at [lat, lon] (m >= 0: cosine family, m < 0: sine family)."""
  x = np.sin(np.deg2rad(np.asarray(lat_deg, np.float64)))
  am = abs(m)
  norm = np.sqrt((2 * l + 1) / (4 * np.pi) * np.exp(scipy.special.gammaln(l - am + 1) - scipy.special.gammaln(l + am + 1)))
  p = norm * scipy.special.lpmv(am, l, x)
  lam = np.deg2rad(np.asarray(lon_deg, np.float64))
  if m == 0:
    return p[:, None] * np.ones_like(lam)[None, :]
  trig = np.cos(am * lam) if m > 0 else np.sin(am * lam)
  return np.sqrt(2.0) * p[:, None] * trig[None, :]


def spherical_field(coef: np.ndarray, lat_deg, lon_deg) -> np.ndarray:
  """coef [2][L][L][N] (part 0: cosine family m >= 0, part 1: sine family; index [part][m][l][n]) ->
  field [n_lat * n_lon, N], flat spectrum p_l = 1 / L (samplers_utils.py:328-346)."""
  coef = np.asarray(coef, np.float64)
  _, L, _, N = coef.shape
  out = np.zeros((len(lat_deg), len(lon_deg), N))
  for l in range(L):
    per_l = np.sqrt(4 * np.pi) * np.sqrt((1.0 / L) / (2 * l + 1))
    for m in range(0, l + 1):
      out += per_l * real_harmonic(l, m, lat_deg, lon_deg)[:, :, None] * coef[0, m, l][None, None, :]
      if m > 0:
        out += per_l * real_harmonic(l, -m, lat_deg, lon_deg)[:, :, None] * coef[1, m, l][None, None, :]
  return out.reshape(len(lat_deg) * len(lon_deg), N)


def device_field(seed: int, stream: int, lat_deg, lon_deg, columns: int, subset=None) -> np.ndarray:
  """The field gc_noise_draw produces for (seed, stream): [G, columns] (or only the columns listed in `subset`: the
  direct evaluation costs L^2 harmonics per column, so large grids are checked on a few columns of the full draw)."""
  L = len(lon_deg) // 2
  z = philox_normals(2 * L * L * columns, seed, stream).reshape(2, L, L, columns)
  if subset is not None:
    z = z[..., np.asarray(subset)]
  return spherical_field(z, lat_deg, lon_deg)


def dpm_solver_2s_sample_churn(network_fn, cond_feats, noisy_slots, init_noise, sigmas, churn_rates, inflation,
                               noise_fn, skip_dead_call=True):
  """DPM-Solver++2S with stochastic churn.  noise_fn(k) -> k-th unit-variance field [G, B, C_out] drawn
  inside the loop (one per step whose rate is > 0)."""
  dt = init_noise.dtype
  sig = np.asarray(sigmas, dtype=dt)
  x = init_noise * sig[0]
  calls = drawn = 0
  for i in range(len(sig) - 1):
    s, s_next = sig[i], sig[i + 1]
    if churn_rates[i] > 0:                                        # samplers_utils.py:441-452
      s_new = s * dt.type(1.0 + churn_rates[i])
      extra = np.sqrt(np.maximum(s_new ** 2 - s ** 2, 0)) * dt.type(inflation)
      x = x + noise_fn(drawn) * extra
      drawn += 1
      s = s_new
    s_mid = np.sqrt(s * s_next)
    x_den = O.preconditioned_denoise(network_fn, cond_feats, noisy_slots, x, s)
    calls += 1
    a_mid = s_mid / s
    x_mid = a_mid * x + (1 - a_mid) * x_den
    if s_next == 0:
      if not skip_dead_call:
        O.preconditioned_denoise(network_fn, cond_feats, noisy_slots, x_mid, s_mid)
        calls += 1
      x = x_den
      continue
    x_mid_den = O.preconditioned_denoise(network_fn, cond_feats, noisy_slots, x_mid, s_mid)
    calls += 1
    a_next = s_next / s
    x = a_next * x + (1 - a_next) * x_mid_den
  return x, calls, drawn
