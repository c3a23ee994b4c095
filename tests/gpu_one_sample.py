"""One nano 20-step sample (for rocprofv3 kernel traces)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers
from oracle import gencast_oracle as O
gr, dims, params, x, sigma = helpers.tiny_setup(batch=1, mesh_size=4, k_hop=8, latent=256, heads=4, ffw=2048,
                                                layers=16, c_in=262, c_out=82, n_lat=73, n_lon=144)
nd = helpers.make_native(gr, dims, params, 1)
if os.environ.get("GC_FEATURES"):                     # "f16": BASELINE configs[4]'s mode (tools/profile_round.sh)
  nd.set_option("features", os.environ["GC_FEATURES"])
nd.set_noisy_slots(np.arange(180, 262))
nd.upload_cond(x)
nd.upload_noise(np.random.default_rng(2).standard_normal((gr.num_grid_nodes, 1, 82)).astype(np.float32))
sig = O.noise_schedule(80, 0.03, 20, 7).astype(np.float32)
for _ in range(int(os.environ.get("SAMPLES", "2"))):
  st = nd.sample_resident(sig)
print(st)
