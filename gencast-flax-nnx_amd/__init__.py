"""MI355X-native GenCast denoiser + DPM-Solver++2S sampling path.

Python host code -> ctypes -> `libgencast_hip.so` (C ABI, include/gencast_hip.h)
-> hand-written HIP kernels for gfx950.  No PyTorch / JAX / Triton in here.
Import as `gencast_flax_nnx_amd` (alias module at the repo root).

Public surface (mirrors the reference's call contracts):
  Denoiser            gencast/denoiser.py:142-202, gencast/denoisers_base.py:28-52
  Sampler             gencast/dpm_solver_plus_plus_2s.py:21-177
  GenCast             gencast/gencast.py:119-185,282-294  (full_sampling)
  EnsembleSampler     common/rollout.py:78-176 (members sharded one per GPU)
  NaNCleaner          gencast/nan_cleaning.py:27-156
  rollout             common/normalization.py:31-238 (InputsAndResiduals), training/train_helpers.py:485-622
                      (autoregressive_rollout); DeviceRollout keeps the context in HBM
"""
from . import config, datasets, geometry, launch, rollout, synthetic, weights  # noqa: F401
from .config import (DenoiserArchitectureConfig, NoiseConfig, NoiseEncoderConfig,  # noqa: F401
                     SamplerConfig, SparseTransformerConfig, TASK, TaskConfig)
from .denoiser import Denoiser  # noqa: F401
from .ensemble import EnsembleSampler, member_seed, member_shard  # noqa: F401
from .gencast import GenCast, create_gencast_model  # noqa: F401
from .nan_cleaning import NaNCleaner  # noqa: F401
from .rollout import DeviceRollout, InputsAndResiduals, autoregressive_rollout  # noqa: F401
from .sampler import Sampler, noise_schedule, stochastic_churn_rate_schedule  # noqa: F401

__all__ = ["Denoiser", "Sampler", "GenCast", "EnsembleSampler", "create_gencast_model",
           "noise_schedule", "config", "datasets", "geometry", "synthetic", "weights", "rollout",
           "InputsAndResiduals", "autoregressive_rollout", "DeviceRollout", "NaNCleaner", "launch"]
