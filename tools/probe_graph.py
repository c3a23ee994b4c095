"""Staged probe of the sampler's HIP-graph replay (progress on stderr, flushed: run under `timeout`).
usage: probe_graph.py [tiny|nano] [levels]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
os.environ.setdefault("GC_TUNE_GRAPH_VERBOSE", "1")
from tests import helpers
from oracle import gencast_oracle as O


def say(*a):
  print(*a, file=sys.stderr, flush=True)


which = sys.argv[1] if len(sys.argv) > 1 else "tiny"
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 4
gr, dims, params, x, sigma = helpers.tiny_setup(batch=1) if which == "tiny" else helpers.nano_setup()
nd = helpers.make_native(gr, dims, params, 1)
nd.set_noisy_slots(np.arange(dims.c_in - dims.c_out, dims.c_in, dtype=np.int32))
nd.upload_cond(x)
nd.upload_noise(np.random.default_rng(2).standard_normal((gr.num_grid_nodes, 1, dims.c_out)).astype(np.float32))
sig = O.noise_schedule(80, 0.03, levels, 7).astype(np.float32)
outs = []
for i in range(4):
  t = time.perf_counter()
  nd.sample_resident(sig, want_stats=False)
  t1 = time.perf_counter()
  say(f"sample {i}: enqueue returned after {1e3 * (t1 - t):.2f} ms")
  nd.sync()
  say(f"sample {i}: synced after {1e3 * (time.perf_counter() - t):.2f} ms; captures {nd.counter('graph_captures')} replays {nd.counter('graph_replays')}")
  outs.append(nd.download_sample())
say("bit-identical across eager / capture / replay:", all(np.array_equal(outs[0], o) for o in outs[1:]))
nd.close()
say("done")
