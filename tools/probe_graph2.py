"""Staged probe: sampler graphs on SEVERAL handles (one host thread, then one thread per handle).  Progress on stderr."""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.getcwd())
os.environ.setdefault("GC_TUNE_GRAPH_VERBOSE", "1")
from tests import helpers
from oracle import gencast_oracle as O


def say(*a):
  print(*a, file=sys.stderr, flush=True)


levels = int(sys.argv[1]) if len(sys.argv) > 1 else 20
stage = sys.argv[2] if len(sys.argv) > 2 else "all"
gr, dims, params, x, sigma = helpers.nano_setup()
sig = O.noise_schedule(80, 0.03, levels, 7).astype(np.float32)


def make(seed):
  nd = helpers.make_native(gr, dims, params, 1)
  nd.set_noisy_slots(np.arange(dims.c_in - dims.c_out, dims.c_in, dtype=np.int32))
  nd.upload_cond(x)
  nd.upload_noise(np.random.default_rng(seed).standard_normal((gr.num_grid_nodes, 1, dims.c_out)).astype(np.float32))
  return nd


a = make(1)
for i in range(3):
  a.sample_resident(sig, want_stats=False)
  a.sync()
say("handle A: eager, captured, replayed", a.counter("graph_captures"), a.counter("graph_replays"))
b = make(2)
say("handle B created while A holds a graph")
b.sample_resident(sig, want_stats=False); b.sync()
say("B eager done")
a.sample_resident(sig, want_stats=False)
say("A replay enqueued; B captures while A runs")
b.sample_resident(sig, want_stats=False)
say("B capture returned")
a.sync(); b.sync()
say("both synced")
t = time.perf_counter()
for _ in range(4):
  a.sample_resident(sig, want_stats=False)
  b.sample_resident(sig, want_stats=False)
te = time.perf_counter() - t
a.sync(); b.sync()
say(f"interleaved replays from one thread: enqueue {1e3 * te:.2f} ms, total {1e3 * (time.perf_counter() - t):.1f} ms")
if stage != "nothreads":
  def run(h):
    for _ in range(4):
      h.sample_resident(sig, want_stats=False)
  ts = [threading.Thread(target=run, args=(h,)) for h in (a, b)]
  t = time.perf_counter()
  for th in ts: th.start()
  say("threads started")
  for th in ts: th.join()
  say("threads joined")
  a.sync(); b.sync()
  say(f"one thread per handle: total {1e3 * (time.perf_counter() - t):.1f} ms")
a.close(); b.close()
say("done")
