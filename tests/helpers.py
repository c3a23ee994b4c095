"""Shared builders for the parity tests (tiny and nano configurations)."""
import dataclasses

import numpy as np

from gencast_flax_nnx_amd import geometry, weights


def graph_dict(gr):
  return dataclasses.asdict(gr)


def tiny_setup(batch=2, seed=0, mesh_size=2, k_hop=2, latent=128, heads=2, ffw=256, layers=2,
               c_in=20, c_out=6, n_lat=13, n_lon=24):
  lat = np.linspace(-90, 90, n_lat)
  lon = np.arange(n_lon) * (360.0 / n_lon)
  gr = geometry.build_denoiser_graph(grid_lat=lat, grid_lon=lon, mesh_size=mesh_size,
                                     attention_k_hop=k_hop)
  dims = weights.ModelDims(c_in=c_in, c_out=c_out, latent=latent, d_model=latent, num_heads=heads,
                           ffw_hidden=ffw, num_layers=layers)
  params = weights.random_params(dims, seed=3)
  rng = np.random.default_rng(seed)
  x = rng.standard_normal((gr.num_grid_nodes, batch, c_in)).astype(np.float32)
  sigma = np.exp(rng.uniform(np.log(0.03), np.log(80.0), size=batch)).astype(np.float32)
  return gr, dims, params, x, sigma


def make_native(gr, dims, params, batch, device_id=0, precision=None):
  from gencast_flax_nnx_amd import _lib
  nd = _lib.NativeDenoiser(latent_size=dims.latent, d_model=dims.d_model, num_heads=dims.num_heads,
                           ffw_hidden=dims.ffw_hidden, num_layers=dims.num_layers, c_in=dims.c_in,
                           c_out=dims.c_out, batch=batch, device_id=device_id)
  if precision:
    nd.set_option("precision", precision)
  nd.set_graph(gr)
  nd.load_weights(params)
  assert nd.missing_weights() == 0
  nd.finalize()
  return nd
