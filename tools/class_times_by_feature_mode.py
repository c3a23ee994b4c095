import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from tests import helpers
from gencast_flax_nnx_amd.sampler import noise_schedule
sig = noise_schedule(80.0, 0.03, 20, 7.0).astype(np.float32)
which = sys.argv[1] if len(sys.argv) > 1 else "nano"
gr, dims, params, x, sigma = helpers.nano_setup() if which == "nano" else helpers.one_degree_setup()
nd = helpers.make_native(gr, dims, params, 1)
for feat in ("f32", "f16"):
  nd.set_option("features", feat)
  nd.set_noisy_slots(np.arange(dims.c_in - dims.c_out, dims.c_in, dtype=np.int32))
  nd.upload_cond(x)
  nd.upload_noise(np.random.default_rng(0).standard_normal((x.shape[0], 1, dims.c_out), dtype=np.float32))
  nd.sample_resident(sig, want_stats=False); nd.sync()
  f0 = nd.counter("range_fallbacks")
  t = time.perf_counter()
  nd.sample_resident(sig, want_stats=False); nd.sync()
  dt = time.perf_counter() - t
  classes = nd.kernel_classes()
  per = {}
  for i, name in enumerate(classes):
    nd.profile_enable(i)
    nd.sample_resident(sig, want_stats=False)
    n, ms = nd.profile_read()
    per[name] = (n, round(ms / 39, 4))
  nd.profile_enable(-1)
  print(which, feat, f"{39 / dt:.1f} calls/s", "fallbacks during the sample:", nd.counter("range_fallbacks") - f0, "launches/call", nd.counter("launches_per_call"))
  print("  ", {k: v for k, v in per.items() if v[0]})
nd.close()
