"""Per-kernel-class device time of one nano denoiser call (HIP events), plus sample time.

    python tests/gpu_class_timing.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers  # noqa: E402
from oracle import gencast_oracle as O  # noqa: E402


def main():
  gr, dims, params, x, sigma = helpers.tiny_setup(
      batch=1, mesh_size=4, k_hop=8, latent=256, heads=4, ffw=2048, layers=16, c_in=262, c_out=82,
      n_lat=73, n_lon=144)
  nd = helpers.make_native(gr, dims, params, 1)
  nd.set_noisy_slots(np.arange(180, 262))
  rng = np.random.default_rng(2)
  nd.upload_cond(x)
  nd.upload_noise(rng.standard_normal((gr.num_grid_nodes, 1, 82)).astype(np.float32))
  sig = O.noise_schedule(80, 0.03, 20, 7).astype(np.float32)
  for _ in range(2):
    st = nd.sample_resident(sig)
  t = time.time()
  st = nd.sample_resident(sig)
  wall = time.time() - t
  print("sample: calls %d device %.2f ms wall %.2f ms -> %.1f calls/s, %.3f ms/call" % (
      st["denoiser_calls"], st["device_ms"], wall * 1e3, st["denoiser_calls"] / (st["device_ms"] * 1e-3),
      st["device_ms"] / st["denoiser_calls"]))
  flops, byts = nd.algorithmic_work()
  print("algorithmic GF %.1f  GB %.3f -> %.1f TF/s" % (flops / 1e9, byts / 1e9,
        flops / (st["device_ms"] / st["denoiser_calls"] * 1e-3) / 1e12))
  tot = 0
  for i, name in enumerate(nd.kernel_classes()):
    nd.profile_enable(i)
    nd.sample_resident(sig)
    n, ms = nd.profile_read()
    nd.profile_enable(-1)
    per_call = ms / st["denoiser_calls"]
    tot += per_call
    print(f"{name:20s} launches/call {n / st['denoiser_calls']:6.1f}  ms/call {per_call:8.4f}  avg us {1e3 * ms / max(n, 1):8.2f}")
  print("sum of classes ms/call %.4f" % tot)


if __name__ == "__main__":
  main()
