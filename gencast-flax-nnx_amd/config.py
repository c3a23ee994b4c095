"""Configuration dataclasses mirroring the reference's field names and defaults.

NoiseEncoderConfig / SparseTransformerConfig / DenoiserArchitectureConfig:
gencast/denoiser.py:47-139.  SamplerConfig / NoiseConfig: gencast/gencast.py:74-115.
TaskConfig + variable tables: graphcast/graphcast.py:61-143, gencast/gencast.py:48-71.
"""
from __future__ import annotations

import dataclasses
from typing import Optional, Tuple


@dataclasses.dataclass(frozen=True)
class NoiseEncoderConfig:
  apply_log_first: bool = True
  base_period: float = 16.0
  num_frequencies: int = 32
  output_sizes: Tuple[int, int] = (32, 16)


@dataclasses.dataclass
class SparseTransformerConfig:
  attention_k_hop: int
  d_model: int
  num_layers: int = 16
  num_heads: int = 4
  attention_type: str = "triblockdiag_mha"
  mask_type: str = "lazy"
  block_q: int = 1024
  block_kv: int = 512
  block_kv_compute: int = 256
  block_q_dkv: int = 512
  block_kv_dkv: int = 1024
  block_kv_dkv_compute: int = 1024
  ffw_winit_final_mult: float = 0.0
  attn_winit_final_mult: float = 0.0
  ffw_hidden: int = 2048


@dataclasses.dataclass
class DenoiserArchitectureConfig:
  sparse_transformer_config: SparseTransformerConfig
  mesh_size: int
  latent_size: int = 512
  hidden_layers: int = 1
  radius_query_fraction_edge_length: float = 0.6
  norm_conditioning_features: Tuple[str, ...] = ("noise_level_encodings",)
  grid2mesh_aggregate_normalization: Optional[float] = None
  node_output_size: Optional[int] = None


@dataclasses.dataclass(frozen=True)
class SamplerConfig:
  max_noise_level: float = 80.0
  min_noise_level: float = 0.03
  num_noise_levels: int = 20
  rho: float = 7.0
  stochastic_churn_rate: float = 2.5
  churn_min_noise_level: float = 0.75
  churn_max_noise_level: float = float("inf")
  noise_level_inflation_factor: float = 1.05


@dataclasses.dataclass(frozen=True)
class NoiseConfig:
  training_noise_level_rho: float = 7.0
  training_max_noise_level: float = 88.0
  training_min_noise_level: float = 0.02


@dataclasses.dataclass(frozen=True)
class TaskConfig:
  input_variables: Tuple[str, ...]
  target_variables: Tuple[str, ...]
  forcing_variables: Tuple[str, ...]
  pressure_levels: Tuple[int, ...]
  input_duration: str


PRESSURE_LEVELS_WEATHERBENCH_13 = (50, 100, 150, 200, 250, 300, 400, 500, 600, 700, 850, 925, 1000)

TARGET_SURFACE_NO_PRECIP_VARS = (
    "2m_temperature", "mean_sea_level_pressure", "10m_v_component_of_wind",
    "10m_u_component_of_wind")
TARGET_ATMOSPHERIC_VARS = (
    "temperature", "geopotential", "u_component_of_wind", "v_component_of_wind",
    "vertical_velocity", "specific_humidity")
ALL_ATMOSPHERIC_VARS = (
    "potential_vorticity", "specific_rain_water_content", "specific_snow_water_content",
    "geopotential", "temperature", "u_component_of_wind", "v_component_of_wind",
    "specific_humidity", "vertical_velocity", "vorticity", "divergence", "relative_humidity",
    "ozone_mass_mixing_ratio", "specific_cloud_liquid_water_content",
    "specific_cloud_ice_water_content", "fraction_of_cloud_cover")
GENERATED_FORCING_VARS = (
    "year_progress_sin", "year_progress_cos", "day_progress_sin", "day_progress_cos")
STATIC_VARS = ("geopotential_at_surface", "land_sea_mask")

TASK = TaskConfig(
    input_variables=(TARGET_SURFACE_NO_PRECIP_VARS + TARGET_ATMOSPHERIC_VARS
                     + GENERATED_FORCING_VARS + STATIC_VARS),
    target_variables=TARGET_SURFACE_NO_PRECIP_VARS + TARGET_ATMOSPHERIC_VARS,
    forcing_variables=GENERATED_FORCING_VARS,
    pressure_levels=PRESSURE_LEVELS_WEATHERBENCH_13,
    input_duration="24h",
)


def num_outputs(task_config: TaskConfig) -> int:
  """gencast/gencast.py:158-169."""
  n_surface = len(set(task_config.target_variables) - set(ALL_ATMOSPHERIC_VARS))
  n_atmos = len(set(task_config.target_variables) & set(ALL_ATMOSPHERIC_VARS))
  return n_surface + len(task_config.pressure_levels) * n_atmos


def nano_architecture(mesh_size: int = 4, d_model: int = 256, num_layers: int = 16,
                      num_heads: int = 4) -> DenoiserArchitectureConfig:
  """The configuration `create_gencast_model` builds (training/train_helpers.py:94-158;
  nano values training/train.py:130-133)."""
  return DenoiserArchitectureConfig(
      sparse_transformer_config=SparseTransformerConfig(
          attention_k_hop=8, d_model=d_model, num_layers=num_layers, num_heads=num_heads),
      mesh_size=mesh_size, latent_size=d_model, hidden_layers=1,
      radius_query_fraction_edge_length=0.6)
