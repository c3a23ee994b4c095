"""Prints per-stage max-abs error of the HIP path against the float64 oracle.

    python tests/gpu_stage_report.py [tiny|nano]
"""
import sys
import os
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gencast_oracle as O  # noqa: E402
from tests import helpers  # noqa: E402


def main(which="tiny"):
  if which == "tiny":
    gr, dims, params, x, sigma = helpers.tiny_setup(batch=2)
  else:
    gr, dims, params, x, sigma = helpers.tiny_setup(
        batch=1, mesh_size=4, k_hop=8, latent=256, heads=4, ffw=2048, layers=int(os.environ.get("LAYERS", 4)),
        c_in=262, c_out=82, n_lat=73, n_lon=144)
  B = x.shape[1]
  t = time.time()
  nd = helpers.make_native(gr, dims, params, B)
  print("native setup %.2fs" % (time.time() - t), nd.debug_attention_stats())
  gd = helpers.graph_dict(gr)
  t = time.time()
  y_ref, inter = O.denoiser_forward(params, gd, x, sigma, num_layers=dims.num_layers,
                                    num_heads=dims.num_heads, attention="neighbour",
                                    return_intermediates=True)
  print("oracle f64 %.2fs" % (time.time() - t))
  t = time.time()
  y = nd.denoise(x, sigma)
  print("gpu denoise %.4fs" % (time.time() - t))

  def rep(name, got, want):
    want = np.asarray(want).reshape(got.shape)
    err = np.abs(got - want).max()
    print(f"{name:8s} max|err| {err:.3e}   ref std {want.std():.3f}  shape {got.shape}")
  rep("cond", nd.debug_fetch("cond"), inter["cond"])
  for k in ["g0", "m0", "e1", "g1", "m2", "f1", "g2"]:
    rep(k, nd.debug_fetch(k), inter[k])
  rep("y", y.reshape(-1, dims.c_out), y_ref)
  # m1 = transformer input: rerun with 0 layers
  nd.debug_set_layer_limit(0)
  nd.denoise(x, sigma)
  rep("m1", nd.debug_fetch("x"), inter["m1"])
  nd.debug_set_layer_limit(-1)


if __name__ == "__main__":
  main(sys.argv[1] if len(sys.argv) > 1 else "tiny")
