"""debug: fused mesh2grid sum vs the two-launch form (which arrays differ, and which one equals the float32 sum of f1)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers
for latent, heads, batch, prec in ((128, 2, 2, "f16x3"), (256, 4, 1, "f32"), (512, 4, 1, "f16x3")):
  gr, dims, params, x, sigma = helpers.tiny_setup(batch=batch, seed=13, latent=latent, heads=heads, ffw=256, layers=1, mesh_size=3, k_hop=2, n_lat=19, n_lon=36)
  outs = {}
  for fused in ("1", "0"):
    os.environ["GC_TUNE_M2G_FUSE_SUM"] = fused
    nd = helpers.make_native(gr, dims, params, batch, precision=prec)
    y = nd.denoise(x, sigma)
    outs[fused] = dict(y=y, agg2=nd.debug_fetch("agg2"), g2=nd.debug_fetch("g2"), f1=nd.debug_fetch("f1"), g1=nd.debug_fetch("g1"), m2=nd.debug_fetch("m2"))
    nd.close()
  for k in ("m2", "g1", "f1", "agg2", "g2", "y"):
    a, b = outs["1"][k], outs["0"][k]
    print(latent, prec, k, "equal" if np.array_equal(a, b) else "DIFF frac %.4f max %.3e" % ((a != b).mean(), np.abs(a - b).max()))
  f1 = outs["0"]["f1"].reshape(gr.num_grid_nodes, 3, batch, -1)
  s = (f1[:, 0] + f1[:, 1]) + f1[:, 2]
  for tag in ("1", "0"):
    a = outs[tag]["agg2"].reshape(s.shape)
    print("   agg2[fused=%s] vs numpy ((f0+f1)+f2): frac diff %.4f" % (tag, (a != s).mean()))
  rows = np.where((outs["1"]["agg2"] != outs["0"]["agg2"]).reshape(gr.num_grid_nodes * batch, -1).any(axis=1))[0]
  print("   differing agg2 rows:", len(rows), rows[:20])
