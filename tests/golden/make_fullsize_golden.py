"""Generates tests/golden/fullsize.npz: THINNED float64-oracle outputs at BASELINE.json's full sizes.

    python tests/golden/make_fullsize_golden.py [nano_sample] [one_degree] [khop16] [one_degree_f16]

The float64 NumPy oracle needs minutes at these sizes, too slow for the `-m gpu` suite, so its
outputs are frozen here (every k-th row only, to keep the file near 1 MB per case) and the GPU
tests compare the HIP path with them on inputs regenerated from the same seeds:

  nano_sample  BASELINE configs[1]: 20-level DPM-Solver++2S sample (39 denoiser calls) of the nano
               model on the 2.5 deg grid, batch 1 -> sample[::5]
  one_degree   BASELINE configs[3]: ONE denoiser call at 1 deg, mesh 5, latent 512, 4 heads of 128,
               FFW 2048, 16 layers, k_hop 8 -> y[::24], m2[::64]
  khop16       SURVEY.md 8d stress: mesh 5 with k_hop = 16 (799 keys per query), 2 layers, heads of
               128 on the 2.5 deg grid -> y[::5], m2[::16]

  one_degree_f16  BASELINE configs[4]'s arithmetic on configs[3]'s sizes: the same 1 deg call in the
               "fp16 node features" mode (oracle header: feature_dtype=float16, float64 arithmetic between
               the rounding points) -> y[::24], m2[::64]

Like denoiser_tiny.npz these vectors come from the restatement (oracle/gencast_oracle.py), not
from the reference itself (JAX is absent here: SURVEY.md 8c) -- "parity unpinned" still applies.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gencast_oracle as O  # noqa: E402
from tests import helpers  # noqa: E402

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fullsize.npz")


def nano_sample():
  gr, dims, params, x, _ = helpers.nano_setup()
  gd = helpers.graph_dict(gr)
  attn = O.make_attention_fn(gd, "dense")
  slots = np.arange(180, 262)
  noise = np.random.default_rng(2).standard_normal((gr.num_grid_nodes, 1, 82)).astype(np.float32)
  sig = O.noise_schedule(80.0, 0.03, 20, 7.0)
  net = lambda f, s: O.denoiser_forward(params, gd, f, s, num_layers=dims.num_layers,
                                        num_heads=dims.num_heads, attention=attn)
  out, calls = O.dpm_solver_2s_sample(net, x.astype(np.float64), slots, noise.astype(np.float64), sig,
                                      skip_dead_call=True)
  return {"nano_sample_out": out[::5].astype(np.float32), "nano_sample_calls": np.int64(calls),
          "nano_sample_scale": np.float64(np.abs(out).max()), "nano_sample_std": np.float64(out.std()),
          "nano_x_sum": np.float64(x.astype(np.float64).sum()),
          "nano_noise_sum": np.float64(noise.astype(np.float64).sum())}


def one_degree():
  gr, dims, params, x, sigma = helpers.one_degree_setup()
  gd = helpers.graph_dict(gr)
  y, inter = O.denoiser_forward(params, gd, x, sigma, num_layers=dims.num_layers, num_heads=dims.num_heads,
                                attention="neighbour_padded", return_intermediates=True)
  return {"one_degree_y": y[::24].astype(np.float32), "one_degree_m2": inter["m2"][::64].astype(np.float32),
          "one_degree_y_std": np.float64(y.std()), "one_degree_x_sum": np.float64(x.astype(np.float64).sum())}


def one_degree_f16():
  gr, dims, params, x, sigma = helpers.one_degree_setup()
  gd = helpers.graph_dict(gr)
  y, inter = O.denoiser_forward(params, gd, x, sigma, num_layers=dims.num_layers, num_heads=dims.num_heads,
                                attention="neighbour_padded", return_intermediates=True, feature_dtype=np.float16)
  return {"one_degree_f16_y": y[::24].astype(np.float32), "one_degree_f16_m2": inter["m2"][::64].astype(np.float32),
          "one_degree_f16_y_std": np.float64(y.std())}


def khop16():
  gr, dims, params, x, sigma = helpers.khop16_setup()
  gd = helpers.graph_dict(gr)
  y, inter = O.denoiser_forward(params, gd, x, sigma, num_layers=dims.num_layers, num_heads=dims.num_heads,
                                attention="neighbour_padded", return_intermediates=True)
  deg = np.diff(gd["khop_rowptr"])
  return {"khop16_y": y[::5].astype(np.float32), "khop16_m2": inter["m2"][::16].astype(np.float32),
          "khop16_max_degree": np.int64(deg.max()), "khop16_nnz": np.int64(len(gd["khop_cols"])),
          "khop16_x_sum": np.float64(x.astype(np.float64).sum())}


if __name__ == "__main__":
  cases = sys.argv[1:] or ["nano_sample", "one_degree", "khop16"]
  data = dict(np.load(PATH)) if os.path.exists(PATH) else {}
  for c in cases:
    t0 = time.time()
    data.update(globals()[c]())
    print(f"{c}: {time.time() - t0:.1f} s", flush=True)
    np.savez_compressed(PATH, **data)
  print("wrote", PATH, os.path.getsize(PATH), "bytes:", sorted(data))
