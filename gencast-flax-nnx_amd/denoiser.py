"""`Denoiser`: the reference's Denoiser call contract on top of the HIP library.

Mirrors gencast/denoisers_base.py:28-52 and gencast/denoiser.py:142-202 (wrapper)
+ :205-341 (architecture): same constructor configs, same
`__call__(inputs, noisy_targets, noise_levels, forcings=None)`, same errors
(`ValueError("noise_levels expected to be shape (batch,).")`, AssertionError when
the data width changes after the first call), lazy graph/weight initialisation on
first call.  All arithmetic happens in libgencast_hip.so; this file only does the
Dataset <-> [grid_node, batch, channel] packing of denoiser.py:770-830.
"""
from __future__ import annotations

import dataclasses
from typing import Dict, Optional

import numpy as np

from . import _lib, config as cfg, datasets, geometry, weights


class Denoiser:
  """Callable with the reference `Denoiser` protocol; owns one GPU handle."""

  def __init__(self,
               noise_encoder_config: Optional[cfg.NoiseEncoderConfig],
               denoiser_architecture_config: cfg.DenoiserArchitectureConfig,
               params: Optional[Dict[str, np.ndarray]] = None,
               *,
               rngs=None,
               gpu_mesh=None,
               device_id: int = 0,
               param_seed: int = 3,
               graph=None,
               options: Optional[Dict[str, str]] = None):
    """`params`: name -> array keyed by the NNX paths (weights.param_specs).  When
    None, synthetic N(0, 1/fan_in) weights are drawn on first call (the reference
    draws its own init from `rngs`; there is no network to fetch checkpoints).
    `graph`: a `geometry.DenoiserGraph`, or the path of an .npz of index arrays dumped from the
    reference (`geometry.load_reference_graph`) -- used instead of building the
    graph here, so a checkpoint runs on exactly the edges it was trained with (the reference's
    trimesh tie-breaking for grid points on mesh edges is not reproducible: geometry.count_m2g_ties).
    `options`: `gc_set_option` key/values applied before the weights are laid out
    (e.g. {"precision": "f32"}, {"features": "f16"}).
    `rngs`, `gpu_mesh`: the reference's constructor arguments (gencast/denoiser.py:153-159), accepted so that a
    call site ports unchanged.  `rngs` (an nnx.Rngs-like object with `.params()`, an int or a numpy Generator)
    only seeds the synthetic weights drawn when `params` is None; `gpu_mesh` (the reference's jax device mesh) is
    ignored -- a handle lives on one GPU, ensemble members are sharded over GPUs by `EnsembleSampler`."""
    del gpu_mesh
    if rngs is not None and params is None:
      if isinstance(rngs, (int, np.integer)):
        param_seed = int(rngs)
      elif isinstance(rngs, np.random.Generator):
        param_seed = int(rngs.integers(0, 2 ** 31 - 1))
      elif hasattr(rngs, "params"):
        param_seed = int(datasets.key_words(rngs.params()).astype(np.uint64).sum() % (2 ** 31 - 1))
    self._noise_cfg = noise_encoder_config or cfg.NoiseEncoderConfig()
    if not self._noise_cfg.apply_log_first:
      raise ValueError("only apply_log_first=True is supported (reference default)")
    if tuple(self._noise_cfg.output_sizes)[1] != weights.COND_DIM:
      raise ValueError("noise encoder must end in 16 conditioning channels")
    self._arch = denoiser_architecture_config
    st = self._arch.sparse_transformer_config
    if st.attention_type not in ("triblockdiag_mha", "mha"):
      raise NotImplementedError(f"attention_type {st.attention_type!r} (TPU splash kernel) is out of scope")
    if not 1 <= int(self._arch.hidden_layers) <= 4:
      raise ValueError("hidden_layers must be in 1..4 (the reference trains with 1, train_helpers.py:137)")
    norm = self._arch.grid2mesh_aggregate_normalization
    if norm is not None and not (float(norm) >= 0.0 and np.isfinite(float(norm))):
      raise ValueError("grid2mesh_aggregate_normalization must be a non-negative constant")
    self._params = params
    self._graph_arg = graph
    self._options = dict(options or {})
    if norm:      # the summed grid2mesh edge messages are divided by it (deep_typed_graph_net.py:396-410; denoiser.py:367)
      self._options.setdefault("grid2mesh_aggregate_normalization", repr(float(norm)))
    self._param_seed = param_seed
    self._device_id = device_id
    self._initialized = False
    self.native: Optional[_lib.NativeDenoiser] = None
    self.graph: Optional[geometry.DenoiserGraph] = None
    self.dims: Optional[weights.ModelDims] = None
    self._grid_shape = None
    self._batch = None

  # ------------------------------------------------------------------------------------------
  def _maybe_init(self, grid_feats_shape, lat, lon):
    """gencast/denoiser.py:343-416 (graphs + networks sized from the first inputs)."""
    g, b, c_in = grid_feats_shape
    if self._initialized:
      if c_in != self.dims.c_in:
        raise AssertionError(
            f"Runtime data width changed: expected {self.dims.c_in}, got {c_in}")
      if b != self._batch:
        raise ValueError(f"batch size changed from {self._batch} to {b}; build a new Denoiser")
      return
    st = self._arch.sparse_transformer_config
    if self._arch.node_output_size is None:
      raise ValueError("denoiser_architecture_config.node_output_size must be set "
                       "(GenCast sets it to the number of predicted channels)")
    if isinstance(self._graph_arg, str):
      self.graph = geometry.load_reference_graph(self._graph_arg, grid_lat=lat, grid_lon=lon,
                                                 attention_k_hop=st.attention_k_hop)
    elif self._graph_arg is not None:
      self.graph = self._graph_arg
    else:
      self.graph = geometry.build_denoiser_graph(
          grid_lat=lat, grid_lon=lon, mesh_size=self._arch.mesh_size,
          attention_k_hop=st.attention_k_hop,
          radius_query_fraction_edge_length=self._arch.radius_query_fraction_edge_length)
    if self.graph.num_grid_nodes != g:
      raise ValueError("lat/lon coordinates do not match the data's grid size")
    self.dims = weights.ModelDims(
        c_in=c_in, c_out=int(self._arch.node_output_size), latent=self._arch.latent_size,
        d_model=st.d_model, num_heads=st.num_heads, ffw_hidden=st.ffw_hidden,
        num_layers=st.num_layers, noise_num_frequencies=self._noise_cfg.num_frequencies,
        noise_hidden=int(self._noise_cfg.output_sizes[0]), hidden_layers=int(self._arch.hidden_layers))
    if self._params is None:
      self._params = weights.random_params(self.dims, seed=self._param_seed)
    self.native = _lib.NativeDenoiser(
        latent_size=self.dims.latent, d_model=self.dims.d_model, num_heads=self.dims.num_heads,
        ffw_hidden=self.dims.ffw_hidden, num_layers=self.dims.num_layers, c_in=c_in,
        c_out=self.dims.c_out, batch=b, device_id=self._device_id,
        noise_num_frequencies=self.dims.noise_num_frequencies, noise_hidden=self.dims.noise_hidden,
        noise_base_period=float(self._noise_cfg.base_period), hidden_layers=self.dims.hidden_layers)
    for k, v in self._options.items():
      self.native.set_option(k, v)
    self.native.set_graph(self.graph)
    self.native.load_weights(self._params)
    self.native.finalize()
    self._batch = b
    self._initialized = True

  def member_lanes(self, count: int):
    """`count` EXTRA library handles on the same GPU with the same graph, weights and options as
    `self.native` (created once, kept).  One nano member leaves about a third of an MI355X idle (every
    kernel of the 81-tile mesh is a short launch): independent members enqueued on separate handles,
    i.e. separate HIP streams, overlap on the chip (`EnsembleSampler(concurrent_members=...)`;
    3 members: 1.45x the calls/s of running them one after the other, tools/members_per_gpu.py)."""
    if not self._initialized:
      raise RuntimeError("member_lanes: the denoiser has not been initialised by a first call / init_for")
    lanes = getattr(self, "_lanes", None)
    if lanes is None:
      lanes = self._lanes = []
    while len(lanes) < count:
      nd = _lib.NativeDenoiser(
          latent_size=self.dims.latent, d_model=self.dims.d_model, num_heads=self.dims.num_heads,
          ffw_hidden=self.dims.ffw_hidden, num_layers=self.dims.num_layers, c_in=self.dims.c_in,
          c_out=self.dims.c_out, batch=self._batch, device_id=self._device_id,
          noise_num_frequencies=self.dims.noise_num_frequencies, noise_hidden=self.dims.noise_hidden,
          noise_base_period=float(self._noise_cfg.base_period), hidden_layers=self.dims.hidden_layers)
      for k, v in self._options.items():
        nd.set_option(k, v)
      nd.set_graph(self.graph)
      nd.load_weights(self._params)
      nd.finalize()
      lanes.append(nd)
    return lanes[:count]

  @staticmethod
  def pack_inputs(inputs: datasets.Dataset, forcings: datasets.Dataset):
    """Datasets -> ([G,B,C] float32, (n_lat, n_lon), lat, lon, C_inputs).

    denoiser.py:770-807: stack inputs, stack forcings, concat on channels, lat/lon
    leading, flatten node = lat_i * n_lon + lon_j.
    """
    sizes = dict(forcings.sizes)
    sizes.update(inputs.sizes)
    si = datasets.dataset_to_stacked(inputs, sizes)
    sf = datasets.dataset_to_stacked(forcings, sizes)
    stacked = np.concatenate([si, sf], axis=-1)            # (batch, lat, lon, channels)
    a = np.transpose(stacked, (1, 2, 0, 3))                # lat, lon leading
    n_lat, n_lon = a.shape[0], a.shape[1]
    feats = np.ascontiguousarray(a.reshape((n_lat * n_lon,) + a.shape[2:]), dtype=np.float32)
    coords = dict(forcings.coords)
    coords.update(inputs.coords)
    if "lat" not in coords or "lon" not in coords:
      raise ValueError("inputs must carry 'lat' and 'lon' coordinates")
    return feats, (n_lat, n_lon), coords["lat"], coords["lon"], si.shape[-1]

  @staticmethod
  def unpack_outputs(out: np.ndarray, grid_shape, template: datasets.Dataset) -> datasets.Dataset:
    """[G,B,C_out] -> Dataset shaped like `template` (denoiser.py:809-830)."""
    a = out.reshape(tuple(grid_shape) + out.shape[1:])     # lat, lon, batch, channels
    a = np.transpose(a, (2, 0, 1, 3))                      # restore_leading_axes
    return datasets.stacked_to_dataset(a, template)

  def noisy_slots(self, inputs: datasets.Dataset, forcings: datasets.Dataset,
                  targets_template: datasets.Dataset) -> np.ndarray:
    """Column of grid_feats for every output channel (sorted target order)."""
    n_inputs = sum(n for _, _, n in datasets.channel_layout(inputs))
    merged = forcings.assign(targets_template)
    where = {name: off for name, off, _ in datasets.channel_layout(merged)}
    slots = []
    for name, _, n in datasets.channel_layout(targets_template):
      slots.extend(range(n_inputs + where[name], n_inputs + where[name] + n))
    return np.asarray(slots, dtype=np.int32)

  # ------------------------------------------------------------------------------------------
  def __call__(self, inputs, noisy_targets, noise_levels, forcings=None, **kwargs):
    if kwargs:
      raise TypeError(f"unexpected arguments {sorted(kwargs)}")
    given = (noisy_targets, inputs, forcings)               # xarray in -> xarray out (datasets.like_inputs)
    inputs = datasets.as_dataset(inputs)
    noisy_targets = datasets.as_dataset(noisy_targets)
    forcings = datasets.as_dataset(forcings)
    forcings = forcings.assign(noisy_targets)                          # denoiser.py:184
    nl = getattr(noise_levels, "dims", None)
    sigma = np.asarray(getattr(noise_levels, "data", noise_levels), dtype=np.float32)
    if (nl is not None and tuple(nl) != ("batch",)) or sigma.ndim != 1:
      raise ValueError("noise_levels expected to be shape (batch,).")
    feats, grid_shape, lat, lon, _ = self.pack_inputs(inputs, forcings)
    if sigma.shape[0] != feats.shape[1]:
      raise ValueError("noise_levels expected to be shape (batch,).")
    self._maybe_init(feats.shape, lat, lon)
    self._grid_shape = grid_shape
    out = self.native.denoise(feats, sigma)
    return datasets.like_inputs(self.unpack_outputs(out, grid_shape, noisy_targets), *given)

  # array-level access for samplers -------------------------------------------------------------
  def init_for(self, inputs, targets_template, forcings):
    """Initialises for these datasets and returns (cond_feats, grid_shape, noisy_slots)."""
    inputs = datasets.as_dataset(inputs)
    template = datasets.as_dataset(targets_template)
    forcings = datasets.as_dataset(forcings)
    merged = forcings.assign(datasets.zeros_like(template))
    feats, grid_shape, lat, lon, _ = self.pack_inputs(inputs, merged)
    self._maybe_init(feats.shape, lat, lon)
    self._grid_shape = grid_shape
    slots = self.noisy_slots(inputs, forcings, template)
    if slots.shape[0] != self.dims.c_out:
      raise ValueError(f"targets_template has {slots.shape[0]} channels, model predicts {self.dims.c_out}")
    return feats, grid_shape, slots


def dims_from_arch(arch: cfg.DenoiserArchitectureConfig, c_in: int, c_out: int,
                   noise: Optional[cfg.NoiseEncoderConfig] = None) -> weights.ModelDims:
  noise = noise or cfg.NoiseEncoderConfig()
  st = arch.sparse_transformer_config
  return weights.ModelDims(c_in=c_in, c_out=c_out, latent=arch.latent_size, d_model=st.d_model,
                           num_heads=st.num_heads, ffw_hidden=st.ffw_hidden, num_layers=st.num_layers,
                           noise_num_frequencies=noise.num_frequencies,
                           noise_hidden=int(noise.output_sizes[0]), hidden_layers=int(arch.hidden_layers))


__all__ = ["Denoiser", "dims_from_arch", "dataclasses"]
