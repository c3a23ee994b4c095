#!/usr/bin/env python3
"""Headline benchmark: denoiser-calls/sec of the nano-GenCast DPM-Solver++2S sampler.

    python bench.py --gpus N --steps K --warmup W

One "step" = one 20-noise-level DPM-Solver++2S sample of ONE ensemble member on the 2.5 deg grid
(BASELINE.json configs[1]) = 39 denoiser forwards (the reference also runs a 40th whose result it
discards, gencast/dpm_solver_plus_plus_2s.py:148-153).  Synthetic ERA5-shaped inputs, random-init
weights of the nano architecture; everything is resident in HBM before the timed region.

N > 1: one process per GPU, every rank samples its own member per step (weak scaling) and each step
starts with the one exchange the path has -- the broadcast of the packed conditioning from rank 0,
issued by the LIBRARY (gc_comm_broadcast_cond: ncclBroadcast over xGMI on the handle's stream).  No
collective runs inside the denoiser.  No torch anywhere: `python bench.py --gpus N` spawns its own N
workers (from a parent that never touches the GPU); under `python -m torch.distributed.run` the ranks
it created are used as they are (RANK / LOCAL_RANK / WORLD_SIZE), torch itself is never imported.

Rank 0 prints ONE JSON line.  `roofline` is measured live (HIP events on the library's stream around
the dominant kernel class during the timed steps).  N = 1 adds: `cpu_baseline` (the NumPy oracle on
the host cores, bounded sample), `f32_exact` (the same workload on the exact-f32 MFMA kernels),
`rollout` (30-step autoregressive forecast wall-clock) and `one_degree` (BASELINE configs[3]: 1 deg
grid, full GenCast widths, its own dominant kernel and roofline).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32 matrix peak (v_mfma_f32_32x32x2_f32)
PEAK_F16_MFMA_TFLOPS = 2516.6  # dense fp16/bf16 MFMA peak (16x the f32 rate; "~2.5 PF dense")
# f16x3 mode: one f32-equivalent product = 3 fp16 MFMAs, so its MFMA ceiling in algorithmic FLOPs
PEAK_F16X3_TFLOPS = PEAK_F16_MFMA_TFLOPS / 3.0
PEAK_HBM_GBS = 8000.0
CALLS_PER_STEP = 39


def parse_args():
  p = argparse.ArgumentParser()
  p.add_argument("--gpus", type=int, default=1)
  p.add_argument("--steps", type=int, default=5)
  p.add_argument("--warmup", type=int, default=1)
  p.add_argument("--no-cpu-baseline", action="store_true")
  p.add_argument("--cpu-baseline-calls", type=int, default=6)   # ~12 s of host work
  p.add_argument("--cpu-threads", type=int, default=16)
  p.add_argument("--rollout-steps", type=int, default=30,
                 help="N = 1 only: also time an autoregressive rollout of this many steps (0 = skip)")
  p.add_argument("--no-extras", action="store_true", help="N = 1: skip f32_exact and one_degree")
  p.add_argument("--force-comm", action="store_true",
                 help="N = 1 rehearsal: run the RCCL exchange on a single-rank communicator")
  p.add_argument("--allow-host-broadcast", action="store_true",
                 help="N > 1: if RCCL cannot span all ranks (e.g. two ranks rehearsing on one GPU), time the host-file "
                      "fallback instead of exiting non-zero; the line then says broadcast: host-file")
  p.add_argument("--extra-steps", type=int, default=5, help="timed steps of the N = 1 extra objects (f32_exact, ...)")
  return p.parse_args()


def time_rollout(steps, arch, params, lat, lon, device_id, *, graph=None, options=None, device_noise=False,
                 label="nano 2.5deg"):
  """30-step autoregressive forecast of one member (normalise -> sample -> residual add -> next
  context), conditioning resident on the GPU (gencast-flax-nnx_amd/rollout.py DeviceRollout)."""
  import dataclasses
  import numpy as np
  from gencast_flax_nnx_amd import GenCast, config, datasets, rollout, synthetic
  inp, tgt1, frc1 = synthetic.make_example(lat, lon, batch=1, seed=0)
  rng = np.random.default_rng(1)

  def stretch(ds, nt, random=True):
    out = {}
    for k, v in ds.items():
      shape = list(v.data.shape)
      shape[v.dims.index("time")] = nt
      out[k] = datasets.Variable(v.dims, rng.standard_normal(shape).astype(np.float32) if random
                                 else np.zeros(shape, np.float32))
    return datasets.Dataset(out, ds.coords)

  def stats(lo, hi, center):
    out = {}
    for name in set(config.TASK.input_variables) | set(config.TASK.target_variables):
      if name in config.ALL_ATMOSPHERIC_VARS:
        out[name] = datasets.Variable(("level",), (center + rng.uniform(lo, hi, 13)).astype(np.float32))
      else:
        out[name] = datasets.Variable((), np.float32(center + rng.uniform(lo, hi)))
    return datasets.Dataset(out)

  targets, forcings = stretch(tgt1, steps, random=False), stretch(frc1, steps)   # targets only give the shapes
  sc = config.SamplerConfig(max_noise_level=80.0, min_noise_level=0.03, num_noise_levels=20, rho=7.0,
                            stochastic_churn_rate=0.0)
  gc = GenCast(config.TASK, dataclasses.replace(arch, node_output_size=82), sc, config.NoiseConfig(), None,
               params=params, rngs=1, device_id=device_id, graph=graph, options=options)
  norm = rollout.InputsAndResiduals(gc, stats(0.5, 2.0, 0.0), stats(-1.0, 1.0, 0.0), stats(0.1, 0.5, 0.0))
  dr = rollout.DeviceRollout(gc, norm, device_noise=device_noise)
  dr.run(inp, targets, forcings, 2)                      # warm-up: lazy init + first launches
  t0 = time.perf_counter()
  preds = dr.run(inp, targets, forcings, steps)
  dt = time.perf_counter() - t0
  finite = all(bool(np.isfinite(v.data).all()) for v in preds.data_vars.values())
  fallbacks = gc.denoiser.native.counter("range_fallbacks")
  gc.denoiser.native.close()
  return {"steps": steps, "denoiser_calls": CALLS_PER_STEP * steps, "seconds": round(dt, 3),
          "ms_per_step": round(1e3 * dt / steps, 2),
          "calls_per_sec_end_to_end": round(CALLS_PER_STEP * steps / dt, 1), "finite": finite,
          "range_fallbacks": fallbacks, "options": options or {}, "device_noise": device_noise,
          "what": f"autoregressive forecast of 1 member, {label}: normalise -> 20-level sample -> residual add "
                  "-> next context, conditioning updated on the GPU (gc_rollout_advance); spherical initial noise "
                  + ("synthesised on the GPU (gc_noise_draw)" if device_noise else "drawn on the host and overlapped")
                  + "; includes D2H of every forecast frame"}


def source_hash():
  """sha256 (16 hex digits) over the kernel and host sources of the library: ties a committed profile to a tree."""
  import hashlib
  h = hashlib.sha256()
  csrc = os.path.join(ROOT, "gencast-flax-nnx_amd", "csrc")
  for name in sorted(os.listdir(csrc)):
    if name.endswith((".hip", ".cpp", ".h", ".inc")):
      h.update(name.encode())
      h.update(open(os.path.join(csrc, name), "rb").read())
  return h.hexdigest()[:16]


def library_source_hash():
  """The source hash csrc/build.sh compiled into the library (gc_build_info: "... src:<hash>"), or None."""
  try:
    from gencast_flax_nnx_amd import _lib
    info = _lib.load_library().gc_build_info().decode()
    return info.split("src:")[1].split()[0] if "src:" in info else None
  except Exception:  # pylint: disable=broad-except
    return None


def profile_figures(dominant, mode=None):
  """traffic (memory-side bytes per launch), MFMA-busy fraction and inter-kernel gaps of the dominant kernel class from
  profiles/ (separate rocprofv3 --pmc / --kernel-trace passes), or None each when the profile is of another tree.
  `mode`: None (nano, float32 features) or "one_degree" / "one_degree_fp16_features" / "fp16_features"."""
  out = {"traffic": None, "mfma_busy": None, "inter_kernel_gaps": None, "from_profile": None}
  try:
    meta = json.load(open(os.path.join(ROOT, "profiles", "profile_meta.json")))
    if meta.get("source_hash") != source_hash():
      out["from_profile"] = {"used": False, "why": "profiles/ were collected on other kernel sources (profile_meta.json)"}
      return out
    if library_source_hash() != source_hash():
      out["from_profile"] = {"used": False, "why": "the loaded library was not built from this tree's sources (gc_build_info)"}
      return out
    tag = meta["tag"]
    out["from_profile"] = {"used": True, "tag": tag, "source_hash": meta["source_hash"],
                           "files": [f"profiles/traffic.json", f"profiles/{tag}_pmc_per_kernel.json",
                                     f"profiles/{tag}_kernel_trace_summary.txt"]}
    traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    out["traffic"] = (traffic.get(mode, {}) if mode else traffic).get(dominant)
    suffix = f"_{mode}" if mode else ""
    out["from_profile"]["files"][1] = f"profiles/{tag}{suffix}_pmc_per_kernel.json"
    per = json.load(open(os.path.join(ROOT, "profiles", f"{tag}{suffix}_pmc_per_kernel.json")))
    from tools.pmc_traffic import kernel_class
    rows = [r for name, r in per.items() if kernel_class(name) == dominant and "mfma_busy_frac" in r]
    if rows:                                # launch-weighted share of cycles with the matrix pipe busy, over the class
      out["mfma_busy"] = round(sum(r["mfma_busy_frac"] * r["launches"] for r in rows) / sum(r["launches"] for r in rows), 4)
    import re
    last = open(os.path.join(ROOT, "profiles", f"{tag}_kernel_trace_summary.txt")).read().strip().splitlines()[-1]
    m = re.search(r"launches (\d+)\s+kernel time ([\d.]+) ms\s+inter-kernel gaps < 20 us: (\d+) sum ([\d.]+) ms avg ([\d.]+) us", last)
    if m:
      out["inter_kernel_gaps"] = {"launches": int(m.group(1)), "kernel_ms": float(m.group(2)), "gaps_counted": int(m.group(3)),
                                  "gap_sum_ms": float(m.group(4)), "gap_avg_us": float(m.group(5))}
  except Exception:  # pylint: disable=broad-except
    pass
  return out


def sampling_stride(per_class, dominant):
  """Every `stride`-th launch of the dominant class is bracketed by HIP events in the timed region.  The stride must be
  coprime with the class's launches per call, or the samples would keep hitting the same few kinds of launch (6 fused
  MLPs per call sampled every 8th: only 3 of the 6 shapes, a biased average)."""
  import math
  n = max(1, int(round(per_class[dominant][0] / CALLS_PER_STEP)))
  for s in (8, 9, 7, 11, 13, 5):
    if math.gcd(s, n) == 1:
      return s
  return 1


def class_profile(nd, sigmas, classes):
  """One untimed sample per kernel class with that class bracketed by HIP events."""
  per_class = {}
  for i, name in enumerate(classes):
    nd.profile_enable(i)
    nd.sample_resident(sigmas, skip_dead_call=True, want_stats=False)
    n, ms = nd.profile_read()
    per_class[name] = (n, ms)
  nd.profile_enable(-1)
  return per_class


def roofline_of(nd, graph, dims, per_class, dominant, dom_launches, dom_ms, precision, value_per_gpu):
  flops, byts = nd.algorithmic_work()
  M, D, F = graph.num_mesh_nodes, dims.d_model, dims.ffw_hidden
  nnz = len(graph.khop_cols)
  per_call_launches = {k: v[0] / CALLS_PER_STEP for k, v in per_class.items()}
  alg_flops = {
      "gc_gemm_ffw1": 2.0 * M * D * F, "gc_gemm_ffw2": 2.0 * M * F * D,
      "gc_gemm_qkv": 2.0 * M * D * 3 * D, "gc_gemm_out": 2.0 * M * D * D,
      "gc_attention": 4.0 * nnz * D,
  }
  if per_call_launches.get("gc_gemm_ffw2", 0.0) == 0.0:   # both FFW layers run inside the ffw1-class launch
    alg_flops["gc_gemm_ffw1"] += alg_flops["gc_gemm_ffw2"]
    alg_flops["gc_gemm_ffw2"] = 0.0
  if dominant in alg_flops:
    flop_per_launch = alg_flops[dominant]
  else:  # fused GNN MLPs: average over the launches of one call
    tr = dims.num_layers * sum(alg_flops.values())
    gnn = flops - tr
    if nd.counter("split_edge"):
      # the edge MLPs run with their first layer split by input block (e @ Wa + (n_s @ Wb)[snd] + (n_r @ Wc)[rcv]):
      # the two node blocks are multiplied once per NODE in the gc_gemm_node class, not once per edge here.  The
      # FLOPs this class EXECUTES are counted, not the concatenated form's.
      L, E = dims.latent, len(graph.g2m_senders) + len(graph.m2g_senders)
      gnn -= 2.0 * (2 * L) * L * E
    flop_per_launch = gnn / max(per_call_launches.get(dominant, 1.0), 1.0)
  avg_s = (dom_ms / max(dom_launches, 1)) * 1e-3
  achieved = flop_per_launch / avg_s / 1e12 if avg_s > 0 else 0.0
  peak = PEAK_F16X3_TFLOPS if precision == "f16x3" else PEAK_F32_MFMA_TFLOPS
  return {
      "bound": "mfma", "kernel": dominant, "achieved": round(achieved, 3),
      "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
      "avg_launch_us": round(avg_s * 1e6, 2), "launches_timed": dom_launches,
      "flop_per_launch": flop_per_launch,
      "frac_of_f32_mfma_peak": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
      "whole_call": {"algorithmic_gflop": round(flops / 1e9, 1), "algorithmic_gb": round(byts / 1e9, 3),
                     "tflops": round(flops * value_per_gpu / 1e12, 2),
                     "frac_of_mfma_peak": round(flops * value_per_gpu / 1e12 / peak, 4),
                     "frac_of_f32_mfma_peak": round(flops * value_per_gpu / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                     "hbm_gbs_algorithmic": round(byts * value_per_gpu / 1e9, 1),
                     "frac_of_hbm_peak": round(byts * value_per_gpu / 1e9 / PEAK_HBM_GBS, 4)},
      "class_ms_per_call": {k: round(v[1] / CALLS_PER_STEP, 4) for k, v in per_class.items()},
      "launches_per_call": int(nd.counter("launches_per_call")),
  }


def time_samples(nd, sigmas, steps):
  nd.sync()
  t0 = time.perf_counter()
  for _ in range(steps):
    nd.sample_resident(sigmas, skip_dead_call=True, want_stats=False)
  nd.sync()
  return time.perf_counter() - t0


def one_degree_objects(device_id, precision, rollout_steps, steps=5):
  """BASELINE.json configs[3]: 1 deg grid (181 x 360), mesh 5, latent 512, 4 heads of 128, FFW 2048, 16
  layers, 1 member: calls/s of the same 20-level sampler, dominant kernel and its roofline.  And configs[4]
  (one member of it): a 30-step autoregressive rollout at 1 deg with fp16 node features."""
  import numpy as np
  from gencast_flax_nnx_amd import _lib, config, geometry, weights
  from gencast_flax_nnx_amd.sampler import noise_schedule
  lat = np.arange(-90.0, 90.0 + 1e-9, 1.0)
  lon = np.arange(0.0, 360.0, 1.0)
  graph = geometry.build_denoiser_graph(grid_lat=lat, grid_lon=lon, mesh_size=5, attention_k_hop=8)
  dims = weights.ModelDims(c_in=262, c_out=82, latent=512, d_model=512, num_heads=4, ffw_hidden=2048, num_layers=16)
  params = weights.random_params(dims, seed=3)
  sampling = _one_degree_sampling(device_id, precision, graph, dims, params, steps)
  if sampling is not None:
    # BASELINE.md section 2's attention stress size: the same model with k_hop 16 (36 key chunks per query tile instead of 13)
    try:
      g16 = geometry.build_denoiser_graph(grid_lat=lat, grid_lon=lon, mesh_size=5, attention_k_hop=16)
      sampling["k_hop16"] = _k_hop16_sampling(device_id, g16, params, max(2, steps // 2))
    except Exception as e:  # pylint: disable=broad-except
      sampling["k_hop16"] = {"error": str(e)}
  roll = None
  if rollout_steps > 0:
    arch = config.nano_architecture(mesh_size=5, d_model=512, num_layers=16, num_heads=4)
    roll = time_rollout(rollout_steps, arch, params, lat, lon, device_id, graph=graph, options={"features": "f16"},
                        device_noise=True, label="1deg grid, full widths, fp16 node features (BASELINE configs[4], 1 member)")
  return sampling, roll


def _k_hop16_sampling(device_id, graph, params, steps):
  import numpy as np
  from gencast_flax_nnx_amd import _lib
  from gencast_flax_nnx_amd.sampler import noise_schedule
  nd = _lib.NativeDenoiser(latent_size=512, d_model=512, num_heads=4, ffw_hidden=2048, num_layers=16, c_in=262,
                           c_out=82, batch=1, device_id=device_id)
  try:
    nd.set_graph(graph)
    nd.load_weights(params)
    nd.finalize()
    nd.set_noisy_slots(np.arange(180, 262, dtype=np.int32))
    rng = np.random.default_rng(0)
    nd.upload_cond(rng.standard_normal((graph.num_grid_nodes, 1, 262), dtype=np.float32))
    nd.upload_noise(rng.standard_normal((graph.num_grid_nodes, 1, 82), dtype=np.float32))
    sigmas = noise_schedule(80.0, 0.03, 20, 7.0).astype(np.float32)
    time_samples(nd, sigmas, 1)
    dt = time_samples(nd, sigmas, steps)
    flops, _ = nd.algorithmic_work()
    value = steps * CALLS_PER_STEP / dt
    return {"workload": "the 1deg model with k_hop 16 (mask entries %d)" % len(graph.khop_cols), "value": round(value, 2),
            "unit": "calls/s", "ms_per_call": round(1e3 / value, 3), "steps": steps,
            "algorithmic_tflops": round(flops * value / 1e12, 1), "attention_items": int(nd.counter("attention_items")),
            "finite": bool(np.isfinite(nd.download_sample()).all())}
  finally:
    nd.close()


def _one_degree_sampling(device_id, precision, graph, dims, params, steps=5):
  import numpy as np
  from gencast_flax_nnx_amd import _lib
  from gencast_flax_nnx_amd.sampler import noise_schedule
  nd = _lib.NativeDenoiser(latent_size=512, d_model=512, num_heads=4, ffw_hidden=2048, num_layers=16, c_in=262,
                           c_out=82, batch=1, device_id=device_id)
  try:
    nd.set_graph(graph)
    nd.load_weights(params)
    nd.finalize()
    nd.set_noisy_slots(np.arange(180, 262, dtype=np.int32))
    rng = np.random.default_rng(0)
    nd.upload_cond(rng.standard_normal((graph.num_grid_nodes, 1, 262), dtype=np.float32))
    nd.upload_noise(rng.standard_normal((graph.num_grid_nodes, 1, 82), dtype=np.float32))
    sigmas = noise_schedule(80.0, 0.03, 20, 7.0).astype(np.float32)
    time_samples(nd, sigmas, 1)                            # warm-up
    classes = nd.kernel_classes()
    per_class = class_profile(nd, sigmas, classes)
    dominant = max(per_class, key=lambda k: per_class[k][1])
    nd.profile_set_stride(sampling_stride(per_class, dominant))
    nd.profile_enable(classes.index(dominant))
    dt = time_samples(nd, sigmas, steps)
    dom_launches, dom_ms = nd.profile_read()
    nd.profile_enable(-1)
    nd.profile_set_stride(1)
    value = steps * CALLS_PER_STEP / dt
    smp = nd.download_sample()
    out = {"workload": "1deg grid (181x360), mesh 5, latent 512, 4 heads of 128, FFW 2048, 16 layers, k_hop 8, "
                       "1 member, 20-level DPM-Solver++2S sample (BASELINE configs[3])",
           "value": round(value, 2), "unit": "calls/s", "ms_per_call": round(1e3 / value, 3),
           "steps": steps, "grid_nodes": graph.num_grid_nodes, "mesh_nodes": graph.num_mesh_nodes,
           "finite": bool(np.isfinite(smp).all()), "range_fallbacks": nd.counter("range_fallbacks"),
           # 321 query tiles on 256 CUs: the attention launches run as a work-item list; 0 = plain launch
           "attention_items": int(nd.counter("attention_items")),
           "roofline": dict(roofline_of(nd, graph, dims, per_class, dominant, dom_launches, dom_ms, precision, value),
                            **{k: v for k, v in profile_figures(dominant, "one_degree").items() if k != "inter_kernel_gaps"})}
    out["m2g_fused_sum"] = int(nd.counter("m2g_fused_sum"))
    # the same workload with fp16 node features (activations stored as 2-byte fp16 arrays: BASELINE configs[4]'s mode)
    nd.set_option("features", "f16")
    time_samples(nd, sigmas, 1)
    dt16 = time_samples(nd, sigmas, steps)
    out["fp16_features"] = {"value": round(steps * CALLS_PER_STEP / dt16, 2), "unit": "calls/s", "steps": steps,
                            "fp16_storage": int(nd.counter("fp16_storage")),
                            "finite": bool(np.isfinite(nd.download_sample()).all())}
    # ... and at the reference's LITERAL dtype: the exact-f32 MFMA family (v_mfma_f32_32x32x2_f32 on WF32 images), its own
    # dominant class and roofline against the dense f32 matrix peak (configs[3] in exact f32 was unmeasured)
    if precision == "f16x3":
      nd.set_option("features", "f32")
      nd.set_option("precision", "f32")
      steps32 = max(2, steps // 2)
      time_samples(nd, sigmas, 1)
      per32 = class_profile(nd, sigmas, classes)
      dom32 = max(per32, key=lambda k: per32[k][1])
      nd.profile_set_stride(sampling_stride(per32, dom32))
      nd.profile_enable(classes.index(dom32))
      dt32 = time_samples(nd, sigmas, steps32)
      l32, ms32 = nd.profile_read()
      nd.profile_enable(-1)
      nd.profile_set_stride(1)
      v32 = steps32 * CALLS_PER_STEP / dt32
      out["f32_exact"] = {"value": round(v32, 2), "unit": "calls/s", "ms_per_call": round(1e3 / v32, 3), "steps": steps32,
                          "dtype": "f32 (v_mfma_f32_32x32x2_f32, f32 accumulate and storage)",
                          "finite": bool(np.isfinite(nd.download_sample()).all()),
                          "attention_items": int(nd.counter("attention_items")),
                          "roofline": roofline_of(nd, graph, dims, per32, dom32, l32, ms32, "f32", v32)}
    return out
  finally:
    nd.close()


def parent_main(args):
  """`python bench.py --gpus N` without a launcher: start N workers, relay rank 0's JSON line."""
  from gencast_flax_nnx_amd import launch   # imports no GPU code
  argv = [os.path.abspath(__file__)] + sys.argv[1:]
  # finite: a rank stuck inside a kernel must not keep the parent alive until the driver's own limit (children are
  # terminated by pid, the parent exits non-zero); generous next to the ~2-4 minutes a full bench takes
  code, out = launch.spawn_workers(argv, args.gpus, env_extra={"GC_BENCH_LAUNCHER": "self"},
                                   timeout=float(os.environ.get("GC_BENCH_WORKER_TIMEOUT", "1500")))
  lines = [ln for ln in out.splitlines() if ln.startswith("{")]
  if code != 0 or not lines:
    print(f"[bench] worker launch failed (exit code {code})", file=sys.stderr)
    return code or 1
  sys.stdout.write(lines[-1] + "\n")
  sys.stdout.flush()
  return 0


def main():
  args = parse_args()
  if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
    raise SystemExit(parent_main(args))
  # RCCL / HIP print banners on stdout; the driver wants exactly ONE JSON line there.
  sys.stdout.flush()
  real_stdout = os.dup(1)
  os.dup2(2, 1)

  import numpy as np  # pylint: disable=import-outside-toplevel
  from gencast_flax_nnx_amd import _lib, config, geometry, launch, synthetic, weights  # noqa: E402
  from gencast_flax_nnx_amd.denoiser import Denoiser  # noqa: E402
  from gencast_flax_nnx_amd.sampler import noise_schedule  # noqa: E402

  rank, local_rank, world = launch.world_from_env()
  args.gpus = world
  use_comm = world > 1 or args.force_comm
  device_id = local_rank % max(_lib.device_count(), 1)

  # ---- workload: BASELINE.json configs[1] (nano, 2.5 deg, 1 member per GPU) -------------------
  lat, lon = synthetic.grid_2p5deg()
  arch = config.nano_architecture(mesh_size=4, d_model=256, num_layers=16, num_heads=4)
  st = arch.sparse_transformer_config
  inp, tgt, frc = synthetic.make_example(lat, lon, batch=1, seed=0)
  helper = Denoiser(None, arch)                       # host-side packing only
  cond, grid_shape, _, _, _ = Denoiser.pack_inputs(inp, frc.assign(tgt.map(np.zeros_like)))
  slots = helper.noisy_slots(inp, frc, tgt)
  dims = weights.ModelDims(c_in=cond.shape[-1], c_out=len(slots), latent=arch.latent_size,
                           d_model=st.d_model, num_heads=st.num_heads, ffw_hidden=st.ffw_hidden,
                           num_layers=st.num_layers)
  params = weights.random_params(dims, seed=3)
  graph = geometry.build_denoiser_graph(grid_lat=lat, grid_lon=lon, mesh_size=arch.mesh_size,
                                        attention_k_hop=st.attention_k_hop)
  nd = _lib.NativeDenoiser(latent_size=dims.latent, d_model=dims.d_model, num_heads=dims.num_heads,
                           ffw_hidden=dims.ffw_hidden, num_layers=dims.num_layers, c_in=dims.c_in,
                           c_out=dims.c_out, batch=1, device_id=device_id)
  nd.set_graph(graph)
  nd.load_weights(params)
  nd.finalize()
  nd.set_noisy_slots(slots)
  sigmas = noise_schedule(80.0, 0.03, 20, 7.0).astype(np.float32)
  noise = np.random.default_rng(1000 + rank).standard_normal(
      (graph.num_grid_nodes, 1, dims.c_out), dtype=np.float32)   # member = rank
  nd.upload_noise(noise)

  # ---- the exchange: conditioning lives on rank 0's GPU, every rank needs it ---------------------
  import socket
  gpu_tag = f"{socket.gethostname()}/{_lib.device_pci_bus_id(device_id)}"   # two ranks on one GPU: no RCCL attempt
  # Leaves the process with a NON-ZERO code (on every rank) when world > 1 and the exchange is not RCCL spanning all
  # ranks, unless --allow-host-broadcast: a run on the host fallback must not pass for a measured multi-GPU result.
  ex = launch.open_exchange(nd, rank, world, _lib.comm_unique_id, gpu_tag=gpu_tag,
                            allow_host_broadcast=args.allow_host_broadcast, force=args.force_comm)
  bcast_mode, rdv = ex.mode, ex.rdv
  if rank == 0:
    nd.upload_cond(cond)                              # resident on rank 0 before anything is timed
  step_no = [0]

  def exchange():
    if bcast_mode == "rccl":
      nd.comm_broadcast_cond(0)                       # ncclBroadcast on the handle's stream + re-pack
    elif bcast_mode == "host-file":                   # insurance only: rank 0's conditioning through /tmp
      blob = rdv.broadcast(f"cond{step_no[0]}", lambda: cond.tobytes())
      step_no[0] += 1
      if rank != 0:
        nd.upload_cond(np.frombuffer(blob, np.float32).reshape(cond.shape))

  def allreduce_max(v):
    if bcast_mode == "rccl":
      return nd.comm_allreduce_max(v)
    if bcast_mode == "host-file":
      key = f"red{step_no[0]}"
      step_no[0] += 1
      rdv.put(f"{key}.{rank}", repr(float(v)).encode())
      return max(float(rdv.get(f"{key}.{r}").decode()) for r in range(world))
    return v

  def barrier():
    nd.sync()
    if use_comm:
      allreduce_max(0.0)

  def one_step():
    exchange()
    return nd.sample_resident(sigmas, skip_dead_call=True, want_stats=False)

  exchange()
  nd.sync()

  # ---- pick the dominant kernel class (untimed pre-pass) ---------------------------------------
  classes = nd.kernel_classes()
  for _ in range(max(args.warmup, 1)):
    one_step()
  nd.sync()
  per_class = class_profile(nd, sigmas, classes) if rank == 0 else {}
  dominant = max(per_class, key=lambda k: per_class[k][1]) if per_class else classes[0]
  dom_idx = classes.index(dominant)

  # ---- timed region: exactly K steps between barriers -------------------------------------------
  # Every rank enqueues the timed steps EAGERLY (sampler graphs off): rank 0 brackets the dominant class's launches
  # with HIP events, which a replayed graph cannot carry, and the ranks' per-rank figures must come from one path
  # (ADVICE r3).  Device time is the same either way (graph replay changes host time only: `graph_replay` below).
  graphs_wanted = os.environ.get("GC_TUNE_GRAPH", "1") != "0"      # the library's default, restored after the timed region
  nd.set_option("graphs", "off")
  if rank == 0:
    nd.profile_set_stride(sampling_stride(per_class, dominant) if per_class else 8)   # ~1 launch in 8: keeps the event records out of the way
    nd.profile_enable(dom_idx)
  barrier()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    one_step()
  nd.sync()
  barrier()
  elapsed = time.perf_counter() - t0
  dom_launches, dom_ms = nd.profile_read() if rank == 0 else (0, 0.0)
  if rank == 0:
    nd.profile_enable(-1)
    nd.profile_set_stride(1)
  nd.set_option("graphs", "on" if graphs_wanted else "off")       # (an A/B run with GC_TUNE_GRAPH=0 keeps its extras eager)
  own_elapsed = elapsed
  fastest = elapsed
  if use_comm:
    elapsed = allreduce_max(own_elapsed)              # MAX over ranks
    fastest = -allreduce_max(-own_elapsed)            # MIN over ranks
  # what the exchange costs on its own (outside the timed region): broadcast + re-pack, waited for
  bcast_ms = None
  if use_comm:
    barrier()
    tb = time.perf_counter()
    for _ in range(3):
      exchange()
    nd.sync()
    bcast_ms = allreduce_max((time.perf_counter() - tb) / 3 * 1e3)

  if os.environ.get("GC_BENCH_CHECKSUM") == "1":
    smp = nd.download_sample()
    print(f"[bench rank {rank}] sample checksum {float(np.abs(smp).sum()):.6f} {float(smp.std()):.6f}", file=sys.stderr)

  if rank == 0:
    precision = os.environ.get("GC_PRECISION", "f16x3")
    total_calls = world * args.steps * CALLS_PER_STEP
    value = total_calls / elapsed
    roofline = roofline_of(nd, graph, dims, per_class, dominant, dom_launches, dom_ms, precision, value / world)
    # Figures that need the profiler come from the committed rocprofv3 passes of tools/profile_round.sh -- and only
    # when those passes were made on THIS tree's kernels (profiles/profile_meta.json records the source hash);
    # otherwise they are left out rather than quoted next to numbers they do not belong to.
    prof = profile_figures(dominant)
    roofline["traffic"] = prof["traffic"]
    roofline["mfma_busy"] = prof["mfma_busy"]
    roofline["inter_kernel_gaps"] = prof["inter_kernel_gaps"]
    roofline["from_profile"] = prof["from_profile"]
    roofline["note"] = (
        "achieved = algorithmic f32 FLOPs of one launch / live HIP-event duration. peak: f16x3 mode executes every "
        "product as 3 fp16 MFMAs, so the MFMA ceiling in algorithmic FLOPs is the dense fp16 peak / 3 (2516.6 / 3 "
        "TFLOP/s); f32 mode: the dense f32 matrix peak 157.3. traffic / mfma_busy / inter_kernel_gaps = memory-side bytes "
        "per launch, SQ_VALU_MFMA_BUSY_CYCLES share and device-timeline gaps from the committed rocprofv3 passes of "
        "tools/profile_round.sh (from_profile says which, and that they were made on this tree's kernels)")
    range_fallbacks = nd.counter("range_fallbacks")

    xs = max(1, args.extra_steps)
    # the same workload through the replayed HIP graph of the sample (what a rollout driver runs): host enqueue time
    # per sample falls from ~25 ms to ~0.1 ms, device time does not change
    graph_replay = None
    if world == 1 and not args.no_extras:
      c0, r0 = nd.counter("graph_captures"), nd.counter("graph_replays")
      time_samples(nd, sigmas, 2)                                  # eager, then capture + first replay
      dtg = time_samples(nd, sigmas, xs)
      graph_replay = {"value": round(xs * CALLS_PER_STEP / dtg, 2), "unit": "calls/s", "steps": xs,
                      "graph_captures": nd.counter("graph_captures") - c0, "graph_replays": nd.counter("graph_replays") - r0}
    f32_exact = None
    if world == 1 and not args.no_extras and precision == "f16x3":
      nd.set_option("precision", "f32")
      time_samples(nd, sigmas, 1)
      pc32 = class_profile(nd, sigmas, classes)
      dom32 = max(pc32, key=lambda k: pc32[k][1])
      nd.profile_set_stride(sampling_stride(pc32, dom32))
      nd.profile_enable(classes.index(dom32))
      dt = time_samples(nd, sigmas, xs)
      l32, ms32 = nd.profile_read()
      nd.profile_enable(-1)
      nd.profile_set_stride(1)
      v32 = xs * CALLS_PER_STEP / dt
      r32 = roofline_of(nd, graph, dims, pc32, dom32, l32, ms32, "f32", v32)
      nd.set_option("precision", "f16x3")
      f32_exact = {"value": round(v32, 2), "unit": "calls/s", "steps": xs,
                   "ms_per_step": round(1e3 * dt / xs, 3), "roofline": r32,
                   "what": "same workload on the exact-f32 MFMA kernels (v_mfma_f32_32x32x2_f32, 24-bit products): "
                           "gc_set_option(precision, f32); roofline.peak = the dense f32 matrix peak"}

    fp16_features = None
    if world == 1 and not args.no_extras and precision == "f16x3":
      nd.set_option("features", "f16")                # BASELINE configs[4]'s arithmetic on the configs[1] workload
      nd.upload_cond(cond)
      nd.upload_noise(noise)
      time_samples(nd, sigmas, 1)
      dt = time_samples(nd, sigmas, xs)
      fp16_features = {"value": round(xs * CALLS_PER_STEP / dt, 2), "unit": "calls/s", "steps": xs,
                       "range_fallbacks": nd.counter("range_fallbacks") - range_fallbacks,
                       "fp16_storage": int(nd.counter("fp16_storage")),
                       "what": "same workload with gc_set_option(features, f16): activations rounded to fp16 where produced "
                               "and STORED as 2-byte fp16 arrays in HBM (DESIGN.md 3b), attention 1 MFMA per product, every "
                               "other product 2 (the activation's lo plane is zero)"}
      nd.set_option("features", "f32")
      nd.upload_cond(cond)
      nd.upload_noise(noise)

    members3 = None
    if world == 1 and not args.no_extras and precision == "f16x3":
      # ensemble throughput of ONE GPU: 3 members in flight on 3 handles (= 3 HIP streams); `value` above stays
      # the 1-member figure BASELINE configs[1] names
      lanes = [nd]
      for i in (1, 2):
        ln = _lib.NativeDenoiser(latent_size=dims.latent, d_model=dims.d_model, num_heads=dims.num_heads,
                                 ffw_hidden=dims.ffw_hidden, num_layers=dims.num_layers, c_in=dims.c_in,
                                 c_out=dims.c_out, batch=1, device_id=device_id)
        ln.set_graph(graph)
        ln.load_weights(params)
        ln.finalize()
        ln.set_noisy_slots(slots)
        nd.sync()
        ln.upload_cond_dev(nd.cond_device_ptr()[0])
        ln.upload_noise(np.random.default_rng(2000 + i).standard_normal(noise.shape, dtype=np.float32))
        lanes.append(ln)

      def group(steps):
        for ln in lanes:
          ln.sync()
        t = time.perf_counter()
        for _ in range(steps):
          for ln in lanes:
            ln.sample_resident(sigmas, skip_dead_call=True, want_stats=False)
        for ln in lanes:
          ln.sync()
        return time.perf_counter() - t
      group(1)
      dt = group(4)
      members3 = {"value": round(len(lanes) * 4 * CALLS_PER_STEP / dt, 2), "unit": "calls/s", "members_in_flight": len(lanes),
                  "steps": 4, "what": "3 ensemble members of the same workload in flight on one GPU, one library handle "
                  "(HIP stream) each, enqueued by one host thread: EnsembleSampler(concurrent_members=3); every "
                  "member's sample is bit-identical to its solo run (tests/test_gpu_host_api.py)"}
      for ln in lanes[1:]:
        ln.close()

    cpu = None
    if world == 1 and not args.no_cpu_baseline:
      from oracle import gencast_oracle as O  # the CPU baseline leg is the ONLY oracle use here
      import dataclasses
      gd = dataclasses.asdict(graph)
      attn = O.make_attention_fn(gd, "triblock")
      xin = cond.copy()
      avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
      cores = min(avail, args.cpu_threads)       # a 1-GPU box's CPU share is 16 cores
      from threadpoolctl import threadpool_limits
      with threadpool_limits(limits=cores):
        tcpu = time.perf_counter()
        for i in range(args.cpu_baseline_calls):
          O.denoiser_forward(params, gd, xin, np.array([float(sigmas[i])], np.float32),
                             num_layers=dims.num_layers, num_heads=dims.num_heads, attention=attn,
                             dtype=np.float32)
        tcpu = time.perf_counter() - tcpu
      cpu = {"value": round(args.cpu_baseline_calls / tcpu, 4), "unit": "denoiser-calls/sec",
             "cores": cores, "kind": "port",
             "sample": f"{args.cpu_baseline_calls} float32 denoiser forwards of the same nano 2.5deg workload "
                       "(NumPy/BLAS restatement of the reference, dense tri-block attention), "
                       f"{tcpu:.1f} s wall"}
    if not getattr(nd, "comm_stuck", False):
      nd.close()
    rollout_info = None
    if world == 1 and args.rollout_steps > 0:
      # second half of BASELINE.json's metric: rollout wall-clock (one member, context resident in HBM)
      rollout_info = time_rollout(args.rollout_steps, arch, params, lat, lon, device_id, device_noise=True)
    one_degree = one_degree_rollout = None
    if world == 1 and not args.no_extras:
      one_degree, one_degree_rollout = one_degree_objects(device_id, precision, args.rollout_steps, xs)
    line = {
        "metric": "denoiser-calls/sec", "value": round(value, 2), "unit": "calls/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("f16x3 (every product = 3 fp16 MFMAs on hi/lo-split f32 operands: 22-bit products, f32 accumulate; "
                  "f32 storage)") if precision == "f16x3" else "f32",
        "data": "synthetic",
        "config": {"workload": "nano-GenCast DPM-Solver++2S 20-step sample, 2.5deg grid (73x144), "
                               "1 ensemble member per GPU, batch 1",
                   "denoiser_calls_per_step": CALLS_PER_STEP, "dead_call_skipped": True,
                   "grid_nodes": graph.num_grid_nodes, "mesh_nodes": graph.num_mesh_nodes, "latent": dims.latent,
                   "layers": dims.num_layers, "heads": dims.num_heads, "ffw_hidden": dims.ffw_hidden,
                   "c_in": dims.c_in, "c_out": dims.c_out, "k_hop": st.attention_k_hop,
                   "parallelism": f"ensemble-dp{world}", "broadcast": bcast_mode, "rccl_ranks": ex.rccl_ranks,
                   "broadcast_bytes_per_step": int(cond.nbytes) if use_comm else 0,
                   "broadcast_ms_per_step": None if bcast_ms is None else round(bcast_ms, 4),
                   "per_rank_calls_per_sec": {"min": round(args.steps * CALLS_PER_STEP / elapsed, 2),
                                              "max": round(args.steps * CALLS_PER_STEP / fastest, 2)},
                   "precision": precision,
                   "timed_path": "eager launches on every rank (sampler graphs off; rank 0 brackets the dominant class with HIP events)",
                   "graphs_outside_timed_region": "on" if graphs_wanted else "off (GC_TUNE_GRAPH=0)",
                   "library_sources": library_source_hash(), "library_built_from_this_tree": library_source_hash() == source_hash(),
                   "launcher": os.environ.get("GC_BENCH_LAUNCHER", "env" if "WORLD_SIZE" in os.environ else "single"),
                   "torch_imported": "torch" in sys.modules},
        "sample_seconds": round(elapsed / args.steps, 4),
        "launches_per_call": roofline["launches_per_call"], "range_fallbacks": range_fallbacks,
        "roofline": roofline, "cpu_baseline": cpu, "f32_exact": f32_exact, "fp16_features": fp16_features, "three_members_in_flight": members3, "graph_replay": graph_replay, "rollout": rollout_info,
        "one_degree": one_degree, "one_degree_rollout_fp16_features": one_degree_rollout,
    }
    if cpu:
      line["gpu_over_cpu"] = round(value / cpu["value"], 1)
    sys.stdout.flush()
    os.write(real_stdout, (json.dumps(line) + "\n").encode())
  elif not getattr(nd, "comm_stuck", False):
    nd.close()
  # two-phase exit barrier (nobody deletes the rendezvous while another rank still reads it); a rank whose helper
  # thread is still blocked inside RCCL leaves with a non-zero code, never 0
  launch.close_exchange(ex)


if __name__ == "__main__":
  main()
