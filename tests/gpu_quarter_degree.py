"""0.25 deg / mesh 6 / full-width configuration (the operational GenCast size; beyond BASELINE.json's configs):
does the library hold up at G = 721 x 1440 = 1 038 240 grid nodes, M = 40 962 mesh nodes, 3.1 M mesh->grid edges?

    python tests/gpu_quarter_degree.py [layers]
Checks: finite, bit-reproducible output; per-class times.  (Index arithmetic beyond 2^31 elements is what this
script exists to exercise: E2 x 512 = 1.6 G values per edge array.)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gencast_flax_nnx_amd import _lib, geometry, weights  # noqa: E402


def main(layers=16):
  lat = np.arange(-90.0, 90.0 + 1e-9, 0.25)
  lon = np.arange(0.0, 360.0, 0.25)
  t = time.time()
  gr = geometry.build_denoiser_graph(grid_lat=lat, grid_lon=lon, mesh_size=6, attention_k_hop=8)
  print("graph %.1fs: G %d M %d E1 %d E2 %d nnz %d" % (time.time() - t, gr.num_grid_nodes, gr.num_mesh_nodes,
                                                       len(gr.g2m_senders), len(gr.m2g_senders), len(gr.khop_cols)), flush=True)
  dims = weights.ModelDims(c_in=262, c_out=82, latent=512, d_model=512, num_heads=4, ffw_hidden=2048,
                           num_layers=layers)
  params = weights.random_params(dims, seed=3)
  t = time.time()
  nd = _lib.NativeDenoiser(latent_size=512, d_model=512, num_heads=4, ffw_hidden=2048, num_layers=layers,
                           c_in=262, c_out=82, batch=1)
  nd.set_graph(gr)
  nd.load_weights(params)
  nd.finalize()
  print("native setup %.1fs" % (time.time() - t), nd.debug_attention_stats(), flush=True)
  if os.environ.get("GC_FEATURES") in ("f16", "f32"):
    nd.set_option("features", os.environ["GC_FEATURES"])
  rng = np.random.default_rng(0)
  x = rng.standard_normal((gr.num_grid_nodes, 1, 262)).astype(np.float32)
  y1 = nd.denoise(x, np.array([3.0], np.float32))
  y2 = nd.denoise(x, np.array([3.0], np.float32))
  assert np.isfinite(y1).all() and np.array_equal(y1, y2)
  print("output std %.3f (finite, bit-reproducible)" % y1.std(), flush=True)
  nd.set_noisy_slots(np.arange(180, 262))
  nd.upload_cond(x)
  nd.upload_noise(rng.standard_normal((gr.num_grid_nodes, 1, 82)).astype(np.float32))
  sig = np.array([80.0, 5.0, 0.0], np.float32)     # 2 levels = 3 denoiser calls
  nd.sample_resident(sig)
  st = nd.sample_resident(sig)
  flops, byts = nd.algorithmic_work()
  ms = st["device_ms"] / st["denoiser_calls"]
  print("0.25deg: %.2f ms/call -> %.2f calls/s; algorithmic %.1f GF %.2f GB -> %.1f TF/s" % (
      ms, 1e3 / ms, flops / 1e9, byts / 1e9, flops / ms / 1e9), flush=True)
  for i, name in enumerate(nd.kernel_classes()):
    nd.profile_enable(i)
    nd.sample_resident(sig)
    n, tot = nd.profile_read()
    nd.profile_enable(-1)
    if n:
      print(f"  {name:18s} {tot / st['denoiser_calls']:8.3f} ms/call  ({n / st['denoiser_calls']:.0f} launches)")
  nd.close()


if __name__ == "__main__":
  main(int(sys.argv[1]) if len(sys.argv) > 1 else 16)
