"""How much of one MI355X does ONE nano member leave idle?  Runs the bench workload (20-level sample, 39
denoiser calls) as (a) 1 handle, batch 1 (the BASELINE configs[1] shape), (b) K handles on K streams of the
same GPU, one member each, enqueued from one host thread, (c) one handle with batch K.  Prints calls/s of
each (a call = one member's denoiser forward, so a batch-K forward counts K).

  python tools/members_per_gpu.py [K ...]        # default 2 3 4
"""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gencast_flax_nnx_amd import _lib, config, geometry, synthetic, weights  # noqa: E402
from gencast_flax_nnx_amd.denoiser import Denoiser  # noqa: E402
from gencast_flax_nnx_amd.sampler import noise_schedule  # noqa: E402

CALLS = 39


def make(batch, dims, graph, params, slots, cond, seed):
  nd = _lib.NativeDenoiser(latent_size=dims.latent, d_model=dims.d_model, num_heads=dims.num_heads,
                           ffw_hidden=dims.ffw_hidden, num_layers=dims.num_layers, c_in=dims.c_in,
                           c_out=dims.c_out, batch=batch, device_id=0)
  nd.set_graph(graph)
  nd.load_weights(params)
  nd.finalize()
  nd.set_noisy_slots(slots)
  nd.upload_cond(np.ascontiguousarray(np.repeat(cond, batch, axis=1)))
  nd.upload_noise(np.random.default_rng(seed).standard_normal((graph.num_grid_nodes, batch, dims.c_out), dtype=np.float32))
  return nd


def timed(handles, sigmas, steps, threads=False, enqueue_time=None):
  for h in handles:
    h.sample_resident(sigmas, skip_dead_call=True, want_stats=False)
  for h in handles:
    h.sync()
  t0 = time.perf_counter()
  if threads:                                   # one enqueueing host thread per handle (ctypes drops the GIL)
    def run(h):
      for _ in range(steps):
        h.sample_resident(sigmas, skip_dead_call=True, want_stats=False)
    ts = [threading.Thread(target=run, args=(h,)) for h in handles]
    for t in ts:
      t.start()
    for t in ts:
      t.join()
  else:
    for _ in range(steps):
      for h in handles:
        h.sample_resident(sigmas, skip_dead_call=True, want_stats=False)
  t1 = time.perf_counter()
  for h in handles:
    h.sync()
  if enqueue_time is not None:
    enqueue_time.append(t1 - t0)
  return time.perf_counter() - t0


def main():
  ks = [int(a) for a in sys.argv[1:]] or [2, 3, 4]
  lat, lon = synthetic.grid_2p5deg()
  arch = config.nano_architecture(mesh_size=4, d_model=256, num_layers=16, num_heads=4)
  st = arch.sparse_transformer_config
  inp, tgt, frc = synthetic.make_example(lat, lon, batch=1, seed=0)
  helper = Denoiser(None, arch)
  cond, _, _, _, _ = Denoiser.pack_inputs(inp, frc.assign(tgt.map(np.zeros_like)))
  slots = helper.noisy_slots(inp, frc, tgt)
  dims = weights.ModelDims(c_in=cond.shape[-1], c_out=len(slots), latent=arch.latent_size, d_model=st.d_model,
                           num_heads=st.num_heads, ffw_hidden=st.ffw_hidden, num_layers=st.num_layers)
  params = weights.random_params(dims, seed=3)
  graph = geometry.build_denoiser_graph(grid_lat=lat, grid_lon=lon, mesh_size=arch.mesh_size,
                                        attention_k_hop=st.attention_k_hop)
  sigmas = noise_schedule(80.0, 0.03, 20, 7.0).astype(np.float32)
  steps = 8
  out = {}
  one = make(1, dims, graph, params, slots, cond, 1)
  enq = []
  dt = timed([one], sigmas, steps, enqueue_time=enq)
  out["1 handle, batch 1"] = round(steps * CALLS / dt, 1)
  out["host enqueue time of one sample (ms) / its GPU time (ms)"] = [round(1e3 * enq[0] / steps, 2), round(1e3 * dt / steps, 2)]
  ref = one.download_sample()
  for k in ks:
    print(f"[members] k = {k}", file=sys.stderr, flush=True)
    hs = [one] + [make(1, dims, graph, params, slots, cond, 1 + i) for i in range(1, k)]
    enq = []
    dt = timed(hs, sigmas, steps, enqueue_time=enq)
    out[f"{k} handles x batch 1"] = round(k * steps * CALLS / dt, 1)
    out[f"{k} handles: host enqueue share of the wall time"] = round(enq[0] / dt, 3)
    print(f"[members] k = {k}: one thread done", file=sys.stderr, flush=True)
    if os.environ.get("MEMBERS_NO_THREADS") != "1":
      dt = timed(hs, sigmas, steps, threads=True)
      print(f"[members] k = {k}: threads done", file=sys.stderr, flush=True)
      out[f"{k} handles x batch 1, one host thread per handle"] = round(k * steps * CALLS / dt, 1)
    same = bool(np.array_equal(hs[0].download_sample(), ref))
    out[f"{k} handles: member 0 bit-identical to the solo run"] = same
    for h in hs[1:]:
      h.close()
    try:
      b = make(k, dims, graph, params, slots, cond, 1)
      print(f"[members] batch {k} handle made", file=sys.stderr, flush=True)
      dt = timed([b], sigmas, steps)
      print(f"[members] batch {k} timed", file=sys.stderr, flush=True)
      out[f"1 handle, batch {k}"] = round(k * steps * CALLS / dt, 1)
      b.close()
    except Exception as e:  # pylint: disable=broad-except
      out[f"1 handle, batch {k}"] = repr(e)
  out["graph captures / replays of handle 0"] = [one.counter("graph_captures"), one.counter("graph_replays")]
  # graph replay vs eager launches on the SAME handle: bit-identical samples
  one.set_option("graphs", "off")
  one.sample_resident(sigmas, skip_dead_call=True, want_stats=False)
  out["eager sample bit-identical to the replayed one"] = bool(np.array_equal(one.download_sample(), ref))
  one.close()
  print(json.dumps(out, indent=1))


if __name__ == "__main__":
  main()
