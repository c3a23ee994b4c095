"""Isotropic white noise on the sphere.

Restates gencast/samplers_utils.py:250-346: `spherical_white_noise_like` draws, per (batch, time,
level) slice of every variable, a Gaussian field with a FLAT power spectrum over total wavenumbers
l = 0 .. n_lon/2 - 1 and unit marginal variance,

    noise(theta, phi) = sqrt(4 pi) * sum_l sqrt(p_l / (2 l + 1)) * sum_{m=-l..l} c_lm Y_lm(theta, phi),
    p_l = 1 / L,  L = n_lon // 2,  c_lm ~ N(0, 1) i.i.d.,

with Y_lm the real orthonormal spherical harmonics (the reference delegates the synthesis to
dinosaur's `RealSphericalHarmonics` grid, `equiangular_with_poles` nodes = the model's lat/lon grid,
samplers_utils.py:60-118; its sqrt(4 pi) factor at :318 is the unit-sphere normalisation of that
basis).  By the addition theorem the marginal variance is sum_l p_l = 1 at every node, poles included,
and the covariance between two nodes depends only on their angular distance gamma:
C(gamma) = sum_l p_l P_l(cos gamma).  dinosaur is not available where this was written, so parity with
the reference's exact coefficient ordering is unpinned; the statistical contract above is what
tests/test_noise.py checks (the reference itself can only promise the spectrum "in expectation").

The synthesis is factorised: a Legendre step per longitudinal wavenumber m, then a Fourier step over
longitude (two small dense products), about 0.3 GFLOP per 82-channel field at 2.5 degrees.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import datasets
from .datasets import Dataset, Variable

try:  # the two products are skinny (K = 72): OpenBLAS is 5-10x SLOWER on them with many threads
  from threadpoolctl import threadpool_limits as _blas_limits
except Exception:  # pylint: disable=broad-except
  import contextlib

  def _blas_limits(limits=None):  # noqa: D401
    return contextlib.nullcontext()


def _normalized_legendre(x: np.ndarray, lmax: int) -> np.ndarray:
  """P[m, l, i] = N_lm P_l^m(x_i) for 0 <= m <= l < lmax (orthonormal on the unit sphere), else 0."""
  x = np.asarray(x, np.float64)
  s = np.sqrt(np.maximum(0.0, 1.0 - x * x))
  P = np.zeros((lmax, lmax, x.shape[0]), np.float64)
  pmm = np.full_like(x, np.sqrt(1.0 / (4.0 * np.pi)))
  for m in range(lmax):
    if m > 0:
      pmm = -np.sqrt((2.0 * m + 1.0) / (2.0 * m)) * s * pmm
    P[m, m] = pmm
    if m + 1 < lmax:
      P[m, m + 1] = np.sqrt(2.0 * m + 3.0) * x * pmm
    for l in range(m + 2, lmax):
      a = np.sqrt((4.0 * l * l - 1.0) / (l * l - m * m))
      b = np.sqrt(((l - 1.0) ** 2 - m * m) / (4.0 * (l - 1.0) ** 2 - 1.0))
      P[m, l] = a * (x * P[m, l - 1] - b * P[m, l - 2])
  return P


class SphericalNoise:
  """Sampler of unit-variance isotropic white noise on a lat/lon grid (degrees)."""

  def __init__(self, lat: np.ndarray, lon: np.ndarray):
    lat = np.asarray(lat, np.float64)
    lon = np.asarray(lon, np.float64)
    if not np.all(np.diff(lat) > 0):
      raise ValueError("Latitude values are expected to be sorted.")           # samplers_utils.py:121-124
    self.n_lat, self.n_lon = lat.shape[0], lon.shape[0]
    self.lmax = max(1, self.n_lon // 2)                                        # wavenumbers 0 .. n_lon/2 - 1 (:336)
    L = self.lmax
    P = _normalized_legendre(np.sin(np.deg2rad(lat)), L)                       # [m, l, lat]
    l = np.arange(L, dtype=np.float64)
    per_l = np.sqrt((1.0 / L) / (2.0 * l + 1.0)) * np.sqrt(4.0 * np.pi)        # :312-318
    self._leg = np.ascontiguousarray(np.transpose(P * per_l[None, :, None], (0, 2, 1)), np.float32)  # [m, lat, l]
    phi = np.deg2rad(lon)
    m = np.arange(L, dtype=np.float64)
    amp = np.where(m == 0, 1.0, np.sqrt(2.0))
    self._cos = (np.cos(phi[:, None] * m[None, :]) * amp[None, :]).astype(np.float32)    # [lon, m]
    self._sin = (np.sin(phi[:, None] * m[None, :]) * amp[None, :]).astype(np.float32)    # [lon, m] (m = 0: zeros)
    self._mask = (m[:, None] <= l[None, :]).astype(np.float32)                 # [m, l]: |m| <= l (:300-303)

  def device_tables(self):
    """(legendre [L, n_lat, L], cos [n_lon, L], sin [n_lon, L]) for `gc_noise_set_tables`: the same
    tables `synthesize` uses (the |m| <= l mask is already in the Legendre table: P_l^m = 0 for l < m,
    and sin(0) = 0 removes the m = 0 sine term)."""
    return (np.ascontiguousarray(self._leg * np.transpose(self._mask, (0, 1))[:, None, :]),
            np.ascontiguousarray(self._cos), np.ascontiguousarray(self._sin))

  @property
  def num_coefficients(self) -> int:
    """Independent normals per field: sum_l (2 l + 1) = L^2."""
    return self.lmax * self.lmax

  def synthesize(self, c_cos: np.ndarray, c_sin: np.ndarray) -> np.ndarray:
    """Coefficients [m, l, C] (cos part: m >= 0, sin part: m >= 1) -> field [lat, lon, C]."""
    C = c_cos.shape[-1]
    L = self.lmax
    with _blas_limits(limits=1):
      f_cos = np.matmul(self._leg, c_cos * self._mask[:, :, None])      # Legendre step: [m, lat, C] (batched GEMM)
      f_sin = np.matmul(self._leg, c_sin * self._mask[:, :, None])
      out = (self._cos @ f_cos.reshape(L, -1)) + (self._sin @ f_sin.reshape(L, -1))   # Fourier step: [lon, lat*C]
    return np.transpose(out.reshape(self.n_lon, self.n_lat, C), (1, 0, 2))

  def sample(self, rng: np.random.Generator, channels: int) -> np.ndarray:
    """[lat, lon, channels] float32, independent fields along the last axis."""
    L = self.lmax
    c_cos = rng.standard_normal((L, L, channels), dtype=np.float32)
    c_sin = rng.standard_normal((L, L, channels), dtype=np.float32)
    c_sin[0] = 0.0                                                             # no sine term at m = 0
    return self.synthesize(c_cos, c_sin).astype(np.float32)

  def covariance(self, cos_gamma: np.ndarray) -> np.ndarray:
    """Theoretical covariance at angular distance gamma: sum_l p_l P_l(cos gamma)."""
    from numpy.polynomial import legendre
    coef = np.full(self.lmax, 1.0 / self.lmax)
    return legendre.legval(np.asarray(cos_gamma, np.float64), coef)


def spherical_white_noise_like(template: Dataset, rngs, generator: Optional[SphericalNoise] = None) -> Dataset:
  """samplers_utils.py:328-346: every variable gets independent fields over its non-(lat, lon) dims."""
  template = datasets.as_dataset(template)
  rng = rngs if isinstance(rngs, np.random.Generator) else np.random.default_rng(rngs)
  gen = generator or SphericalNoise(template.coords["lat"], template.coords["lon"])
  out = {}
  for name, v in template.items():
    if "lat" not in v.dims or "lon" not in v.dims:
      raise ValueError(f"{name}: spherical noise needs 'lat' and 'lon' dimensions")
    other = [d for d in v.dims if d not in ("lat", "lon")]
    n = int(np.prod([v.sizes[d] for d in other])) if other else 1
    field = gen.sample(rng, n)                                                 # [lat, lon, n]
    field = field.reshape([v.sizes["lat"], v.sizes["lon"]] + [v.sizes[d] for d in other])
    order = [(["lat", "lon"] + other).index(d) for d in v.dims]
    out[name] = Variable(v.dims, np.ascontiguousarray(np.transpose(field, order)).astype(v.data.dtype))
  return Dataset(out, template.coords)


def packed_noise(gen: SphericalNoise, rng: np.random.Generator, batch: int, channels: int) -> np.ndarray:
  """[G, B, C] initial noise for `gc_sample` / `Sampler(init_noise=...)`, node = lat_i * n_lon + lon_j."""
  f = gen.sample(rng, batch * channels)
  return np.ascontiguousarray(f.reshape(gen.n_lat * gen.n_lon, batch, channels))
