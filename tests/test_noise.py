"""Spherical white noise: the statistical contract of
gencast/samplers_utils.py:250-346 -- unit marginal variance everywhere (poles included), isotropy
(covariance = sum_l p_l P_l(cos gamma)), flat spectrum over l = 0 .. n_lon/2 - 1."""
import numpy as np

from gencast_flax_nnx_amd import noise, synthetic
from gencast_flax_nnx_amd.datasets import Dataset, Variable


def _gen(n_lat=19, n_lon=36):
  lat = np.linspace(-90, 90, n_lat)
  lon = np.arange(n_lon) * (360.0 / n_lon)
  return noise.SphericalNoise(lat, lon), lat, lon


def test_basis_is_orthonormal_under_quadrature():
  """The normalised Legendre table integrates to delta_ll' (Gauss-Legendre quadrature in x)."""
  L = 12
  x, w = np.polynomial.legendre.leggauss(64)
  P = noise._normalized_legendre(x, L)
  for m in (0, 1, 5, 11):
    gram = 2 * np.pi * np.einsum("li,ki,i->lk", P[m, m:], P[m, m:], w) * (1.0 if m == 0 else 1.0)
    np.testing.assert_allclose(gram, np.eye(L - m), atol=1e-10)


def test_unit_variance_everywhere_and_isotropic_covariance():
  gen, lat, lon = _gen()
  rng = np.random.default_rng(0)
  n = 6000
  f = gen.sample(rng, n).astype(np.float64)                   # [lat, lon, n]
  var = (f * f).mean(-1)
  assert abs(f.mean()) < 5e-3
  np.testing.assert_allclose(var, 1.0, atol=0.08)             # incl. both poles (rows 0 and -1)
  np.testing.assert_allclose(var[[0, -1]].mean(), 1.0, atol=0.04)
  # covariance between node pairs depends only on the angular distance
  pts = [(3, 0), (3, 9), (9, 4), (15, 20), (0, 0), (18, 7), (12, 30)]
  def xyz(i, j):
    th, ph = np.deg2rad(lat[i]), np.deg2rad(lon[j])
    return np.array([np.cos(th) * np.cos(ph), np.cos(th) * np.sin(ph), np.sin(th)])
  for a in pts:
    for b in pts:
      cg = float(np.clip(xyz(*a) @ xyz(*b), -1, 1))
      emp = (f[a[0], a[1]] * f[b[0], b[1]]).mean()
      assert abs(emp - gen.covariance(cg)) < 0.06, (a, b, emp, gen.covariance(cg))


def test_flat_spectrum_through_the_exact_synthesis():
  """One unit coefficient at (l, m) gives a field of variance 4 pi p_l / (2 l + 1) * <Y_lm^2> -- summed over
  m at fixed l the node-wise power is p_l exactly (addition theorem)."""
  gen, lat, lon = _gen(13, 24)
  L = gen.lmax
  for l in (0, 1, 5, L - 1):
    power = np.zeros((13, 24))
    for m in range(0, l + 1):
      for part in ("cos", "sin"):
        if part == "sin" and m == 0:
          continue
        cc = np.zeros((L, L, 1), np.float32)
        cs = np.zeros((L, L, 1), np.float32)
        (cc if part == "cos" else cs)[m, l, 0] = 1.0
        power += gen.synthesize(cc, cs)[..., 0].astype(np.float64) ** 2
    np.testing.assert_allclose(power, 1.0 / L, rtol=2e-5)
  assert gen.num_coefficients == L * L


def test_dataset_interface_and_packing():
  lat, lon = np.linspace(-90, 90, 7), np.arange(12) * 30.0
  _, tgt, _ = synthetic.make_example(lat, lon, batch=2, seed=0)
  out = noise.spherical_white_noise_like(tgt, 3)
  for k, v in tgt.items():
    assert out[k].dims == v.dims and out[k].data.shape == v.data.shape and out[k].data.dtype == v.data.dtype
  again = noise.spherical_white_noise_like(tgt, 3)
  for k in tgt.keys():
    np.testing.assert_array_equal(out[k].data, again[k].data)          # deterministic in the seed
  gen = noise.SphericalNoise(lat, lon)
  p = noise.packed_noise(gen, np.random.default_rng(1), 2, 5)
  assert p.shape == (7 * 12, 2, 5) and p.dtype == np.float32
  bad = Dataset({"x": Variable(("batch", "time"), np.zeros((1, 1), np.float32))}, dict(lat=lat, lon=lon))
  try:
    noise.spherical_white_noise_like(bad, 0)
    assert False
  except ValueError:
    pass
