"""1 deg / full-width configuration (BASELINE.json configs[3]): timing + sanity on the GPU.

    python tests/gpu_one_degree.py [layers [k_hop]]      (k_hop 16: BASELINE.md's attention stress size)
G = 181x360 = 65160 grid nodes, mesh 5 (10242 nodes), latent = d_model = 512, 4 heads of 128,
ffw 2048, k-hop 8.  Parity at this size: tests/test_gpu_parity.py::test_one_degree_16_layers_matches_oracle_fixture
(thinned float64-oracle fixture); this script only times the kernels.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gencast_flax_nnx_amd import _lib, geometry, weights  # noqa: E402


def main(layers=16, k_hop=8):
  lat = np.arange(-90.0, 90.0 + 1e-9, 1.0)
  lon = np.arange(0.0, 360.0, 1.0)
  t = time.time()
  gr = geometry.build_denoiser_graph(grid_lat=lat, grid_lon=lon, mesh_size=5, attention_k_hop=k_hop)
  print("graph %.1fs: G %d M %d E1 %d E2 %d nnz %d" % (time.time() - t, gr.num_grid_nodes, gr.num_mesh_nodes,
                                                       len(gr.g2m_senders), len(gr.m2g_senders), len(gr.khop_cols)))
  dims = weights.ModelDims(c_in=262, c_out=82, latent=512, d_model=512, num_heads=4, ffw_hidden=2048,
                           num_layers=layers)
  params = weights.random_params(dims, seed=3)
  t = time.time()
  nd = _lib.NativeDenoiser(latent_size=512, d_model=512, num_heads=4, ffw_hidden=2048, num_layers=layers,
                           c_in=262, c_out=82, batch=1)
  nd.set_graph(gr)
  nd.load_weights(params)
  nd.finalize()
  print("native setup %.1fs" % (time.time() - t), nd.debug_attention_stats())
  if os.environ.get("GC_FEATURES") in ("f16", "f32"):   # "f16": BASELINE configs[4]'s arithmetic (tools/profile_round.sh)
    nd.set_option("features", os.environ["GC_FEATURES"])
  rng = np.random.default_rng(0)
  x = rng.standard_normal((gr.num_grid_nodes, 1, 262)).astype(np.float32)
  y1 = nd.denoise(x, np.array([3.0], np.float32))
  y2 = nd.denoise(x, np.array([3.0], np.float32))
  assert np.isfinite(y1).all() and np.array_equal(y1, y2)
  print("output std %.3f (finite, bit-reproducible)" % y1.std())
  nd.set_noisy_slots(np.arange(180, 262))
  nd.upload_cond(x)
  nd.upload_noise(rng.standard_normal((gr.num_grid_nodes, 1, 82)).astype(np.float32))
  sig = np.array([80.0, 20.0, 5.0, 1.0, 0.0], np.float32)     # 4 levels = 7 denoiser calls
  nd.sample_resident(sig)
  st = nd.sample_resident(sig)
  flops, byts = nd.algorithmic_work()
  ms = st["device_ms"] / st["denoiser_calls"]
  print("1deg: %.2f ms/call -> %.1f calls/s; algorithmic %.1f GF %.2f GB -> %.1f TF/s" % (
      ms, 1e3 / ms, flops / 1e9, byts / 1e9, flops / ms / 1e9))
  if os.environ.get("ONE_DEGREE_QUICK") == "1":       # counter passes: the calls above are enough
    return
  for i, name in enumerate(nd.kernel_classes()):
    nd.profile_enable(i)
    nd.sample_resident(sig)
    n, tot = nd.profile_read()
    nd.profile_enable(-1)
    if n:
      print(f"  {name:18s} {tot / st['denoiser_calls']:8.3f} ms/call  ({n / st['denoiser_calls']:.0f} launches)")


if __name__ == "__main__":
  main(int(sys.argv[1]) if len(sys.argv) > 1 else 16, int(sys.argv[2]) if len(sys.argv) > 2 else 8)
