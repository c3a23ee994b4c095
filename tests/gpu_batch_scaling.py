"""Throughput with several ensemble members batched on ONE GPU (members > GPUs are
batched along B).  Prints member-calls/s = B * denoiser calls/s for B = 1, 2, 4 at the nano size."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gencast_oracle as O  # noqa: E402
from tests import helpers  # noqa: E402

for B in (1, 2, 4):
  gr, dims, params, x, sigma = helpers.tiny_setup(batch=B, mesh_size=4, k_hop=8, latent=256, heads=4, ffw=2048,
                                                  layers=16, c_in=262, c_out=82, n_lat=73, n_lon=144)
  nd = helpers.make_native(gr, dims, params, B)
  nd.set_noisy_slots(np.arange(180, 262))
  nd.upload_cond(x)
  nd.upload_noise(np.random.default_rng(2).standard_normal((gr.num_grid_nodes, B, 82)).astype(np.float32))
  sig = O.noise_schedule(80, 0.03, 20, 7).astype(np.float32)
  nd.sample_resident(sig)
  st = nd.sample_resident(sig)
  ms = st["device_ms"] / st["denoiser_calls"]
  out = nd.download_sample()
  print(f"B={B}: {ms:.3f} ms per batched call -> {1e3 / ms:.1f} calls/s, {B * 1e3 / ms:.1f} member-calls/s; "
        f"finite {bool(np.isfinite(out).all())}")
  nd.close()
