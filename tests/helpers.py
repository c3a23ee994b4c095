"""Shared builders for the parity tests (tiny and nano configurations)."""
import dataclasses

import numpy as np

from gencast_flax_nnx_amd import geometry, weights


def graph_dict(gr):
  return dataclasses.asdict(gr)


def tiny_setup(batch=2, seed=0, mesh_size=2, k_hop=2, latent=128, heads=2, ffw=256, layers=2,
               c_in=20, c_out=6, n_lat=13, n_lon=24, hidden_layers=1):
  lat = np.linspace(-90, 90, n_lat)
  lon = np.arange(n_lon) * (360.0 / n_lon)
  gr = geometry.build_denoiser_graph(grid_lat=lat, grid_lon=lon, mesh_size=mesh_size,
                                     attention_k_hop=k_hop)
  dims = weights.ModelDims(c_in=c_in, c_out=c_out, latent=latent, d_model=latent, num_heads=heads,
                           ffw_hidden=ffw, num_layers=layers, hidden_layers=hidden_layers)
  params = weights.random_params(dims, seed=3)
  rng = np.random.default_rng(seed)
  x = rng.standard_normal((gr.num_grid_nodes, batch, c_in)).astype(np.float32)
  sigma = np.exp(rng.uniform(np.log(0.03), np.log(80.0), size=batch)).astype(np.float32)
  return gr, dims, params, x, sigma


def make_native(gr, dims, params, batch, device_id=0, precision=None):
  from gencast_flax_nnx_amd import _lib
  nd = _lib.NativeDenoiser(latent_size=dims.latent, d_model=dims.d_model, num_heads=dims.num_heads,
                           ffw_hidden=dims.ffw_hidden, num_layers=dims.num_layers, c_in=dims.c_in,
                           c_out=dims.c_out, batch=batch, device_id=device_id, hidden_layers=dims.hidden_layers)
  if precision:
    nd.set_option("precision", precision)
  nd.set_graph(gr)
  nd.load_weights(params)
  assert nd.missing_weights() == 0
  nd.finalize()
  return nd


def nano_setup(batch=1):
  """BASELINE.json configs[1]: nano model on the 2.5 deg grid."""
  return tiny_setup(batch=batch, mesh_size=4, k_hop=8, latent=256, heads=4, ffw=2048, layers=16, c_in=262,
                    c_out=82, n_lat=73, n_lon=144)


def one_degree_setup(layers=16, k_hop=8):
  """BASELINE.json configs[3]: 1 deg grid, mesh 5, full GenCast widths (latent 512, 4 heads of 128)."""
  lat = np.arange(-90.0, 90.0 + 1e-9, 1.0)
  lon = np.arange(0.0, 360.0, 1.0)
  gr = geometry.build_denoiser_graph(grid_lat=lat, grid_lon=lon, mesh_size=5, attention_k_hop=k_hop)
  dims = weights.ModelDims(c_in=262, c_out=82, latent=512, d_model=512, num_heads=4, ffw_hidden=2048,
                           num_layers=layers)
  params = weights.random_params(dims, seed=3)
  x = np.random.default_rng(0).standard_normal((gr.num_grid_nodes, 1, 262)).astype(np.float32)
  return gr, dims, params, x, np.array([3.0], np.float32)


def khop16_setup():
  """SURVEY.md 8d stress case: mesh 5 with k_hop = 16 (up to 799 keys per query), heads of 128."""
  return tiny_setup(batch=1, seed=5, mesh_size=5, k_hop=16, latent=256, heads=2, ffw=256, layers=2, c_in=20,
                    c_out=6, n_lat=73, n_lon=144)
