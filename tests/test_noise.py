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


def test_philox_known_answers_and_device_layout():
  """The counter-based generator of csrc/gc_noise.hip, restated in oracle/noise_oracle.py, against the
  published Philox4x32-10 known-answer vectors (Random123 kat_vectors), and the Box-Muller layout."""
  from oracle import noise_oracle as NO
  kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
         ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
         ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
          (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
  for ctr, key, want in kat:
    got = NO.philox4x32_10(np.array(ctr, np.uint32), key)
    assert tuple(int(x) for x in got) == want
  z = NO.philox_normals(200003, seed=99, stream=3)
  assert z.shape == (200003,) and abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
  np.testing.assert_array_equal(z[:1000], NO.philox_normals(1000, 99, 3))          # a prefix is a prefix
  assert not np.array_equal(z[:1000], NO.philox_normals(1000, 99, 4))              # streams differ
  assert not np.array_equal(z[:1000], NO.philox_normals(1000, 98, 3))              # seeds differ


def test_oracle_direct_synthesis_matches_the_product_tables():
  """oracle.spherical_field evaluates the real spherical harmonics directly (scipy lpmv); the product's
  factorised Legendre / Fourier tables (what the device kernels consume) must give the same field."""
  from oracle import noise_oracle as NO
  gen, lat, lon = _gen(13, 24)
  L = gen.lmax
  coef = NO.philox_normals(2 * L * L * 5, 1, 0).reshape(2, L, L, 5)
  direct = NO.spherical_field(coef, lat, lon)
  prod = gen.synthesize(coef[0].copy(), coef[1].copy()).reshape(-1, 5)
  np.testing.assert_allclose(prod, direct, atol=2e-6)
  leg, ct, st = gen.device_tables()
  assert leg.shape == (L, 13, L) and ct.shape == (24, L) and st.shape == (24, L)
  assert np.all(leg[3, :, :3] == 0) and np.all(st[:, 0] == 0)                      # l < m and the m = 0 sine term
  # the device's two steps, written out with the tables
  f = np.einsum("mal,pmln->pman", leg.astype(np.float64), coef.astype(np.float64))
  x = np.einsum("om,man->aon", ct.astype(np.float64), f[0]) + np.einsum("om,man->aon", st.astype(np.float64), f[1])
  np.testing.assert_allclose(x.reshape(-1, 5), direct, atol=2e-6)
