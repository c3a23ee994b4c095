"""`GenCast`: the Predictor-like wrapper the rollout / evaluation harness calls.

Mirrors gencast/gencast.py:119-185,282-294: constructor signature
(task_config, denoiser_architecture_config, sampler_config, noise_config,
noise_encoder_config), `num_outputs` inference (:158-169), the sampler singleton
(:182-185), `__call__` and `full_sampling(inputs, targets_template, forcings)`.

Known reference defect NOT copied: `GenCast.__call__` forwards `forcings`
positionally into the denoiser's `noise_levels` slot (gencast.py:282-287 vs
denoiser.py:172-178).  Here `__call__` takes `noise_levels` explicitly.
`loss` / `loss_and_predictions` are training-only and out of scope.
"""
from __future__ import annotations

import dataclasses
from typing import Dict, Optional

import numpy as np

from . import config as cfg
from .denoiser import Denoiser
from .sampler import Sampler


class GenCast:

  def __init__(self,
               task_config: cfg.TaskConfig,
               denoiser_architecture_config: cfg.DenoiserArchitectureConfig,
               sampler_config: Optional[cfg.SamplerConfig] = None,
               noise_config: Optional[cfg.NoiseConfig] = None,
               noise_encoder_config: Optional[cfg.NoiseEncoderConfig] = None,
               gpu_mesh=None,
               rngs=0,
               *,
               params: Optional[Dict[str, np.ndarray]] = None,
               device_id: int = 0,
               graph=None,
               options: Optional[Dict[str, str]] = None):
    """Positional order as gencast/gencast.py:145-154 (`..., noise_encoder_config, gpu_mesh, rngs`).  `gpu_mesh`
    (the reference's jax device mesh) is accepted and ignored: one GenCast drives one GPU, ensemble members are
    spread over GPUs by `EnsembleSampler` / `launch.py`.  `rngs`: an int seed, a numpy Generator, or an
    nnx.Rngs-like object whose `.noise()` returns key words (kept as it is: the sampler draws its keys from it)."""
    del gpu_mesh
    if isinstance(rngs, np.random.Generator) or hasattr(rngs, "noise"):
      self.rngs = rngs
    else:
      self.rngs = np.random.default_rng(rngs)
    denoiser_architecture_config = dataclasses.replace(
        denoiser_architecture_config, node_output_size=cfg.num_outputs(task_config))
    self.denoiser = Denoiser(noise_encoder_config, denoiser_architecture_config, params,
                             device_id=device_id, graph=graph, options=options)
    self._sampler_config = sampler_config or cfg.SamplerConfig()
    self._noise_config = noise_config
    self._sampler = Sampler(self.denoiser, **dataclasses.asdict(self._sampler_config))

  def __call__(self, inputs, targets_template, noise_levels, forcings=None, **kwargs):
    """One raw denoiser evaluation (`targets_template` plays the noisy targets)."""
    return self.denoiser(inputs, targets_template, noise_levels, forcings, **kwargs)

  def full_sampling(self, inputs, targets_template, forcings=None, **kwargs):
    """gencast/gencast.py:289-294."""
    return self._sampler(inputs, targets_template, forcings, rngs=self.rngs, **kwargs)

  def as_predictor_fn(self):
    """`PredictorFn(rng, inputs, targets_template, forcings)` (common/rollout.py:29-38): the functional form
    `chunked_prediction_generator` calls (:345-349).  `rng`: a numpy Generator, an int seed, or an object
    with `.noise()` returning key words (the reference passes a jax PRNG key / nnx.Rngs stream)."""
    def predictor_fn(rng, inputs, targets_template, forcings, **optional_kwargs):
      return self._sampler(inputs, targets_template, forcings, rngs=rng, **optional_kwargs)
    return predictor_fn

  def loss(self, *args, **kwargs):
    raise NotImplementedError("training (loss) is outside the sampling hot path")

  loss_and_predictions = loss


def create_gencast_model(task_config: cfg.TaskConfig = cfg.TASK, *, mesh_size: int = 3,
                         d_model: int = 256, num_layers: int = 2, num_heads: int = 4,
                         stochastic_churn_rate: float = 0.0, params=None, rngs=0,
                         device_id: int = 0) -> GenCast:
  """training/train_helpers.py:94-158 (defaults included)."""
  sampler_config = cfg.SamplerConfig(max_noise_level=80.0, min_noise_level=0.03,
                                     num_noise_levels=20, rho=7.0,
                                     stochastic_churn_rate=float(stochastic_churn_rate))
  arch = cfg.nano_architecture(mesh_size=mesh_size, d_model=d_model, num_layers=num_layers,
                               num_heads=num_heads)
  return GenCast(task_config, arch, sampler_config, cfg.NoiseConfig(), None, params=params,
                 rngs=rngs, device_id=device_id)
