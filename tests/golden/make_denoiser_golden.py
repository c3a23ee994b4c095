"""Generates tests/golden/denoiser_tiny.npz: float64 oracle outputs for a tiny config.

    python tests/golden/make_denoiser_golden.py

The reference holds no golden vectors for this path and cannot run here
(SURVEY.md 8c), so these vectors come from the NumPy restatement
(oracle/gencast_oracle.py).  They pin the oracle against accidental change and
give the GPU tests a fixture that does not need the oracle at run time.
Inputs are regenerated from seeds by tests/helpers.py::tiny_setup.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gencast_oracle as O  # noqa: E402
from tests import helpers  # noqa: E402

gr, dims, params, x, sigma = helpers.tiny_setup(batch=2)
gd = helpers.graph_dict(gr)
y, inter = O.denoiser_forward(params, gd, x, sigma, num_layers=dims.num_layers,
                              num_heads=dims.num_heads, attention="neighbour",
                              return_intermediates=True)
# sampler: 6-level schedule, batch 2
sig = O.noise_schedule(80.0, 0.03, 6, 7.0)
slots = np.arange(dims.c_in - dims.c_out, dims.c_in)
rng = np.random.default_rng(5)
noise = rng.standard_normal((gr.num_grid_nodes, 2, dims.c_out))
net = lambda f, s: O.denoiser_forward(params, gd, f, s, num_layers=dims.num_layers,
                                      num_heads=dims.num_heads, attention="neighbour")
sample, calls = O.dpm_solver_2s_sample(net, x.astype(np.float64), slots, noise, sig)
# inputs (x, noise) are regenerated from seeds by the tests; their sums guard the seeds
out = dict(x_sum=np.float64(x.astype(np.float64).sum()), sigma=sigma, y=y, cond=inter["cond"],
           sampler_sigmas=sig, sampler_noise_sum=np.float64(noise.sum()), sampler_slots=slots,
           sampler_out=sample, sampler_calls=np.int64(calls))
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "denoiser_tiny.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes; calls", calls, "y std", y.std(), "sample std", sample.std())
