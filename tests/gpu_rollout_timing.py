"""Rollout wall-clock on the GPU (BASELINE.json: "+ rollout wall-clock"): a 30-step autoregressive
forecast of ONE nano member at 2.5 deg, conditioning resident in HBM (DeviceRollout) vs the
host-composed loop (re-normalise + re-pack + upload every step).

    python tests/gpu_rollout_timing.py [steps]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gencast_flax_nnx_amd import GenCast, config, datasets, rollout, synthetic, weights  # noqa: E402
from gencast_flax_nnx_amd.denoiser import dims_from_arch  # noqa: E402
from tests.test_rollout import _stats  # noqa: E402


def main(steps=30):
  lat, lon = synthetic.grid_2p5deg()
  arch = config.nano_architecture(mesh_size=4, d_model=256, num_layers=16, num_heads=4)
  inp, tgt1, frc1 = synthetic.make_example(lat, lon, batch=1, seed=0)
  rng = np.random.default_rng(1)

  def stretch(ds, nt):
    out = {}
    for k, v in ds.items():
      shape = list(v.data.shape)
      shape[v.dims.index("time")] = nt
      out[k] = datasets.Variable(v.dims, rng.standard_normal(shape).astype(np.float32))
    return datasets.Dataset(out, ds.coords)

  targets, forcings = stretch(tgt1, steps), stretch(frc1, steps)
  import dataclasses
  arch = dataclasses.replace(arch, node_output_size=82)
  params = weights.random_params(dims_from_arch(arch, 262, 82), seed=3)
  sc = config.SamplerConfig(max_noise_level=80.0, min_noise_level=0.03, num_noise_levels=20, rho=7.0,
                            stochastic_churn_rate=0.0)
  gc = GenCast(config.TASK, arch, sc, config.NoiseConfig(), None, params=params, rngs=1)
  norm = rollout.InputsAndResiduals(gc, *_stats(config.TASK))
  dr = rollout.DeviceRollout(gc, norm)
  dr.run(inp, targets, forcings, 2)                               # warm-up (lazy init, first launches)
  t = time.perf_counter()
  dev = dr.run(inp, targets, forcings, steps)
  t_dev = time.perf_counter() - t
  t = time.perf_counter()
  _, host, _ = rollout.autoregressive_rollout(norm, inp, targets, forcings, steps)
  t_host = time.perf_counter() - t
  calls = 39 * steps
  print(f"rollout {steps} steps x 39 denoiser calls, nano 2.5deg, 1 member, 1 GPU")
  print(f"  device-resident context : {t_dev:7.3f} s  ({1e3 * t_dev / steps:6.1f} ms/step, {calls / t_dev:6.1f} calls/s)"
        f"  median step {np.median(dr.last_step_ms):.1f} ms")
  print(f"  host-composed context   : {t_host:7.3f} s  ({1e3 * t_host / steps:6.1f} ms/step, {calls / t_host:6.1f} calls/s)")
  ok = all(np.isfinite(v.data).all() for v in dev.data_vars.values())
  print("  outputs finite:", ok)
  gc.denoiser.native.close()


if __name__ == "__main__":
  main(int(sys.argv[1]) if len(sys.argv) > 1 else 30)
