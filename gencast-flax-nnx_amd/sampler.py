"""DPM-Solver++2S sampler with the reference `Sampler` call contract.

Mirrors gencast/samplers_base.py:22-44 and
gencast/dpm_solver_plus_plus_2s.py:21-177: `Sampler(denoiser, **SamplerConfig)`,
`__call__(inputs, targets_template, forcings=None, rngs=...)`.  Schedules are the
host functions of gencast/samplers_utils.py:350-431; the 20-step loop itself runs
inside ONE native call (`gc_sample`), so Python is out of the hot loop.

Differences from the reference, on purpose:
  * `stochastic_churn_rate > 0` runs: the reference's loop calls `utils.apply_stochastic_churn_arr`,
    which does not exist (dpm_solver_plus_plus_2s.py:131), so the arithmetic is taken from the Dataset
    version `apply_stochastic_churn` (samplers_utils.py:434-452); the per-step spherical noise is
    synthesised on the device (`gc_set_churn`, Philox streams seeded from `rngs`), which needs an
    equiangular grid with poles like the reference's own generator;
  * the initial noise is the reference's isotropic spherical white noise
    (samplers_utils.py:250-346, restated in noise.py) when the grid is equiangular with poles
    (n_lon == 2 (n_lat - 1)), else white Gaussian per grid node; `noise_kind="white"` forces the
    latter, `init_noise=` supplies any field;
  * the last step's mid-point denoiser evaluation, whose result the reference
    discards (:148-153), is skipped unless `evaluate_dead_call=True`.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import datasets
from .denoiser import Denoiser


def rho_inverse_cdf(min_value, max_value, rho, cdf):
  """Quantiles of the rho distribution (gencast/samplers_utils.py:350-383)."""
  return (min_value ** (1 / rho) + cdf * (max_value ** (1 / rho) - min_value ** (1 / rho))) ** rho


def noise_schedule(max_noise_level: float = 80.0, min_noise_level: float = 0.002,
                   num_noise_levels: int = 30, rho: float = 7.0) -> np.ndarray:
  """Descending noise levels with a trailing zero (samplers_utils.py:395-412)."""
  levels = rho_inverse_cdf(min_value=min_noise_level, max_value=max_noise_level, rho=rho,
                           cdf=np.linspace(1, 0, num_noise_levels))
  return np.append(levels, 0.0)


def stochastic_churn_rate_schedule(noise_levels, stochastic_churn_rate: float = 0.0,
                                   churn_min_noise_level: float = 0.05,
                                   churn_max_noise_level: float = 50.0) -> np.ndarray:
  """Per-step churn rate (samplers_utils.py:415-431)."""
  num = len(noise_levels) - 1
  rate = min(stochastic_churn_rate / num, np.sqrt(2) - 1)
  return ((churn_min_noise_level <= noise_levels[:-1])
          & (noise_levels[:-1] <= churn_max_noise_level)) * rate


def _draw_noise(rngs, shape) -> np.ndarray:
  if rngs is None:
    raise ValueError("Must pass rngs (a numpy Generator, an int seed, or an object with .noise())")
  if isinstance(rngs, np.random.Generator):
    gen = rngs
  elif isinstance(rngs, (int, np.integer)):
    gen = np.random.default_rng(int(rngs))
  elif hasattr(rngs, "noise"):
    key = rngs.noise()
    gen = np.random.default_rng(datasets.key_words(key).tolist())
  else:
    raise TypeError(f"unsupported rngs: {type(rngs)}")
  return gen.standard_normal(shape, dtype=np.float32)


class Sampler:
  """DPM-Solver++2S (gencast/dpm_solver_plus_plus_2s.py:21-45)."""

  def __init__(self, denoiser: Denoiser, max_noise_level: float, min_noise_level: float,
               num_noise_levels: int, rho: float, stochastic_churn_rate: float,
               churn_min_noise_level: float, churn_max_noise_level: float,
               noise_level_inflation_factor: float, *, evaluate_dead_call: bool = False,
               noise_kind: str = "auto", device_noise: bool = False):
    self._noise_levels = noise_schedule(max_noise_level, min_noise_level, num_noise_levels, rho)
    self._stochastic_churn = stochastic_churn_rate > 0.0
    self._per_step_churn_rates = stochastic_churn_rate_schedule(
        self._noise_levels, stochastic_churn_rate, churn_min_noise_level, churn_max_noise_level)
    self._noise_level_inflation_factor = noise_level_inflation_factor
    self._denoiser = denoiser
    self._evaluate_dead_call = evaluate_dead_call
    if noise_kind not in ("auto", "spherical", "white"):
      raise ValueError("noise_kind must be 'auto', 'spherical' or 'white'")
    self.noise_kind = noise_kind
    self.device_noise = device_noise          # draw the initial state on the GPU too (gc_noise_draw)
    self._noise_gen = None
    self._tables_on = None                    # the native handle that holds the noise tables
    self.sigma_data = 1.0
    self.last_stats = None

  @property
  def noise_levels(self) -> np.ndarray:
    return self._noise_levels

  def draw_noise(self, rngs, shape, template) -> np.ndarray:
    """Unit-variance initial noise [G, B, C]: spherical white noise on a qualifying grid (the
    reference's choice, dpm_solver_plus_plus_2s.py:71-78 via samplers_utils.py:328-346), else white."""
    from . import noise as _noise
    lat, lon = template.coords.get("lat"), template.coords.get("lon")
    ok = lat is not None and lon is not None and len(lon) == 2 * (len(lat) - 1) and np.all(np.diff(lat) > 0)
    kind = self.noise_kind
    if kind == "auto":
      kind = "spherical" if ok else "white"
    if kind == "white":
      return _draw_noise(rngs, shape)
    if not ok:
      raise ValueError(f"Unexpected number of longitude nodes. Expected {2 * (len(lat) - 1)}, got {len(lon)}")
    if self._noise_gen is None or (self._noise_gen.n_lat, self._noise_gen.n_lon) != (len(lat), len(lon)):
      self._noise_gen = _noise.SphericalNoise(lat, lon)
    if rngs is None:
      raise ValueError("Must pass rngs (a numpy Generator, an int seed, or an object with .noise())")
    gen = rngs if isinstance(rngs, np.random.Generator) else None
    if gen is None:
      seed = int(rngs) if isinstance(rngs, (int, np.integer)) else \
          datasets.key_words(rngs.noise()).tolist()
      gen = np.random.default_rng(seed)
    return _noise.packed_noise(self._noise_gen, gen, shape[1], shape[2])

  def ensure_device_noise(self, native, template) -> None:
    """Uploads the spherical-harmonic tables of the template's grid to `native` (once per handle)."""
    from . import noise as _noise
    if self._tables_on is native:
      return
    lat, lon = template.coords.get("lat"), template.coords.get("lon")
    if lat is None or lon is None or len(lon) != 2 * (len(lat) - 1):
      n = None if lon is None else len(lon)
      raise ValueError(f"Unexpected number of longitude nodes. Expected {2 * (len(lat) - 1) if lat is not None else '?'}, got {n}")
    if self._noise_gen is None or (self._noise_gen.n_lat, self._noise_gen.n_lon) != (len(lat), len(lon)):
      self._noise_gen = _noise.SphericalNoise(lat, lon)
    native.noise_set_tables(len(lat), len(lon), *self._noise_gen.device_tables())
    self._tables_on = native

  @staticmethod
  def seed_from(rngs) -> int:
    """A 64-bit Philox key from whatever `rngs` is (Generator, int seed, object with `.noise()`)."""
    if rngs is None:
      raise ValueError("Must pass rngs (a numpy Generator, an int seed, or an object with .noise())")
    if isinstance(rngs, np.random.Generator):
      return int(rngs.integers(0, 2 ** 63 - 1))
    if isinstance(rngs, (int, np.integer)):
      return int(rngs) & (2 ** 64 - 1)
    words = datasets.key_words(rngs.noise()).astype(np.uint64)
    return int((int(words[0]) << 32 | int(words[-1])) & (2 ** 64 - 1))

  def __call__(self, inputs, targets_template, forcings=None, rngs=None, *,
               init_noise: Optional[np.ndarray] = None):
    template = datasets.as_dataset(targets_template)
    cond, grid_shape, slots = self._denoiser.init_for(inputs, template, forcings)
    native = self._denoiser.native
    native.set_noisy_slots(slots)
    shape = (cond.shape[0], cond.shape[1], self._denoiser.dims.c_out)
    on_device = init_noise is None and self.device_noise and self.noise_kind != "white"
    if self._stochastic_churn or on_device:
      self.ensure_device_noise(native, template)
      native.noise_seed(self.seed_from(rngs), 0)
    if self._stochastic_churn:                                  # dpm_solver_plus_plus_2s.py:128-137
      native.set_churn(self._per_step_churn_rates, self._noise_level_inflation_factor)
    else:
      native.set_churn(None)
    native.upload_cond(cond)
    if on_device:
      native.noise_draw()                                       # stream 0; churn fields follow from stream 1
    else:
      if init_noise is None:
        init_noise = self.draw_noise(rngs, shape, template)
      init_noise = np.asarray(init_noise, dtype=np.float32)
      if init_noise.shape != shape:
        raise ValueError(f"init_noise must have shape {shape}")
      native.upload_noise(init_noise)
    sigmas = np.asarray(self._noise_levels, dtype=np.float32)   # cast like :66
    self.last_stats = native.sample_resident(sigmas, skip_dead_call=not self._evaluate_dead_call, want_stats=True)
    out = native.download_sample()
    # xarray in -> xarray out: the harness concatenates what this returns with xr.concat (train_helpers.py:569-624)
    return datasets.like_inputs(Denoiser.unpack_outputs(out, grid_shape, template), targets_template, inputs, forcings)
