"""Soak: N samples back to back on one handle (eager, then graph replay), every sample bit-identical to the first;
device time per sample min / median / max.    python tests/gpu_soak.py [samples [nano|one_degree [f32|f16]]]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers  # noqa: E402
from oracle import gencast_oracle as O  # noqa: E402


def main(n=200, size="nano", features="f32"):
  gr, dims, params, x, sigma = helpers.nano_setup() if size == "nano" else helpers.one_degree_setup()
  nd = helpers.make_native(gr, dims, params, 1)
  nd.set_option("features", features)
  nd.set_noisy_slots(np.arange(180, 262))
  nd.upload_cond(x)
  nd.upload_noise(np.random.default_rng(2).standard_normal((gr.num_grid_nodes, 1, 82)).astype(np.float32))
  sig = O.noise_schedule(80, 0.03, 20, 7).astype(np.float32)
  for mode in ("off", "on"):
    nd.set_option("graphs", mode)
    ms, first = [], None
    t0 = time.time()
    for i in range(n):
      st = nd.sample_resident(sig)
      ms.append(st["device_ms"])
      if i % max(1, n // 6) == 0 or i == n - 1:
        out = nd.download_sample()
        assert np.isfinite(out).all()
        if first is None:
          first = out
        assert np.array_equal(out, first), f"sample {i} differs from sample 0 (graphs {mode})"
    ms = np.array(ms[2:])
    print(f"{size} features {features} graphs {mode}: {n} samples in {time.time() - t0:.1f} s; device ms per sample min {ms.min():.2f} median {np.median(ms):.2f} "
          f"max {ms.max():.2f} -> {39e3 / np.median(ms):.1f} calls/s; range fallbacks {nd.counter('range_fallbacks')}, "
          f"attention items {nd.counter('attention_items')}", flush=True)
  nd.close()


if __name__ == "__main__":
  main(int(sys.argv[1]) if len(sys.argv) > 1 else 200, sys.argv[2] if len(sys.argv) > 2 else "nano",
       sys.argv[3] if len(sys.argv) > 3 else "f32")
