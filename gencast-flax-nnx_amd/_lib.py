"""ctypes binding of libgencast_hip.so (include/gencast_hip.h).

There is no CPU fallback: if the shared library is missing this module raises at
import of the native symbols, and without a HIP device `gc_create` fails.
"""
from __future__ import annotations

import ctypes
import os
import sys
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libgencast_hip.so")

GC_OK = 0
GC_ERR_INVALID_ARGUMENT = 1
GC_ERR_NO_DEVICE = 2
GC_ERR_HIP = 3
GC_ERR_STATE = 4
GC_ERR_UNSUPPORTED = 5
GC_ERR_INTERNAL = 6
GC_ERR_COMM = 7
COMM_ID_BYTES = 128


class GcConfig(ctypes.Structure):
  _fields_ = [
      ("latent_size", ctypes.c_int32), ("d_model", ctypes.c_int32), ("num_heads", ctypes.c_int32),
      ("ffw_hidden", ctypes.c_int32), ("num_layers", ctypes.c_int32), ("c_in", ctypes.c_int32),
      ("c_out", ctypes.c_int32), ("batch", ctypes.c_int32),
      ("noise_num_frequencies", ctypes.c_int32), ("noise_hidden", ctypes.c_int32),
      ("noise_base_period", ctypes.c_float)]


class GcSampleStats(ctypes.Structure):
  _fields_ = [("denoiser_calls", ctypes.c_int32), ("device_ms", ctypes.c_float)]


class GencastHipError(RuntimeError):
  """A HIP / state failure inside the native library."""


_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)
_hp = ctypes.c_void_p

# name -> (restype, argtypes); every symbol include/gencast_hip.h (+ _debug.h) declares.
SIGNATURES = {
    "gc_abi_version": (ctypes.c_int, []),
    "gc_build_info": (ctypes.c_char_p, []),
    "gc_device_count": (ctypes.c_int, []),
    "gc_device_pci_bus_id": (ctypes.c_int, [ctypes.c_int32, ctypes.c_char_p, ctypes.c_int64]),
    "gc_last_error": (ctypes.c_char_p, [_hp]),
    "gc_create": (ctypes.c_int, [ctypes.POINTER(GcConfig), ctypes.c_int, ctypes.POINTER(_hp)]),
    "gc_destroy": (None, [_hp]),
    "gc_set_option": (ctypes.c_int, [_hp, ctypes.c_char_p, ctypes.c_char_p]),
    "gc_set_graph": (ctypes.c_int, [_hp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _i32p, _i32p,
                                    ctypes.c_int32, _i32p, _i32p, _i32p, _i32p, _f32p, _f32p, _f32p,
                                    _f32p, _f32p]),
    "gc_load_weight": (ctypes.c_int, [_hp, ctypes.c_char_p, _f32p, _i64p, ctypes.c_int32]),
    "gc_missing_weights": (ctypes.c_int, [_hp, _i32p]),
    "gc_finalize": (ctypes.c_int, [_hp]),
    "gc_denoise": (ctypes.c_int, [_hp, _f32p, _f32p, _f32p]),
    "gc_set_noisy_slots": (ctypes.c_int, [_hp, _i32p]),
    "gc_sample": (ctypes.c_int, [_hp, _f32p, _f32p, _f32p, ctypes.c_int32, ctypes.c_int32, _f32p,
                                 ctypes.POINTER(GcSampleStats)]),
    "gc_upload_cond": (ctypes.c_int, [_hp, _f32p]),
    "gc_upload_cond_dev": (ctypes.c_int, [_hp, ctypes.c_void_p]),
    "gc_upload_noise": (ctypes.c_int, [_hp, _f32p]),
    "gc_sample_resident": (ctypes.c_int, [_hp, _f32p, ctypes.c_int32, ctypes.c_int32,
                                          ctypes.POINTER(GcSampleStats)]),
    "gc_download_sample": (ctypes.c_int, [_hp, _f32p]),
    "gc_stash_sample": (ctypes.c_int, [_hp]),
    "gc_download_stash": (ctypes.c_int, [_hp, _f32p]),
    "gc_rollout_plan": (ctypes.c_int, [_hp, _i32p, _i32p, _i32p, _f32p, _f32p, ctypes.c_int32]),
    "gc_rollout_advance": (ctypes.c_int, [_hp, _f32p]),
    "gc_download_cond": (ctypes.c_int, [_hp, _f32p]),
    "gc_sync": (ctypes.c_int, [_hp]),
    "gc_cond_device_ptr": (ctypes.c_int, [_hp, ctypes.POINTER(ctypes.c_void_p), _i64p]),
    "gc_commit_cond": (ctypes.c_int, [_hp]),
    "gc_num_kernel_classes": (ctypes.c_int, []),
    "gc_kernel_class_name": (ctypes.c_char_p, [ctypes.c_int]),
    "gc_profile_enable": (ctypes.c_int, [_hp, ctypes.c_int]),
    "gc_profile_set_stride": (ctypes.c_int, [_hp, ctypes.c_int]),
    "gc_profile_read": (ctypes.c_int, [_hp, _i32p, _f32p]),
    "gc_algorithmic_work": (ctypes.c_int, [_hp, ctypes.POINTER(ctypes.c_double),
                                           ctypes.POINTER(ctypes.c_double)]),
    "gc_get_counter": (ctypes.c_int, [_hp, ctypes.c_char_p, _i64p]),
    "gc_noise_set_tables": (ctypes.c_int, [_hp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _f32p, _f32p, _f32p]),
    "gc_noise_seed": (ctypes.c_int, [_hp, ctypes.c_uint64, ctypes.c_uint64]),
    "gc_noise_draw": (ctypes.c_int, [_hp]),
    "gc_download_noise": (ctypes.c_int, [_hp, _f32p]),
    "gc_set_churn": (ctypes.c_int, [_hp, _f32p, ctypes.c_int32, ctypes.c_float]),
    "gc_comm_unique_id": (ctypes.c_int, [ctypes.c_void_p]),
    "gc_comm_init": (ctypes.c_int, [_hp, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32]),
    "gc_comm_info": (ctypes.c_int, [_hp, _i32p, _i32p]),
    "gc_comm_broadcast_cond": (ctypes.c_int, [_hp, ctypes.c_int32]),
    "gc_comm_allreduce_max": (ctypes.c_int, [_hp, ctypes.POINTER(ctypes.c_double)]),
    "gc_comm_destroy": (ctypes.c_int, [_hp]),
    # include/gencast_hip_debug.h (tests only)
    "gc_debug_fetch": (ctypes.c_int, [_hp, ctypes.c_char_p, _f32p, ctypes.c_int64, _i64p, _i64p]),
    "gc_debug_set_layer_limit": (ctypes.c_int, [_hp, ctypes.c_int32]),
    "gc_debug_set_stop": (ctypes.c_int, [_hp, ctypes.c_int32, ctypes.c_int32]),
    "gc_debug_mesh_permutation": (ctypes.c_int, [_hp, _i32p]),
    "gc_debug_attention_stats": (ctypes.c_int, [_hp, _i64p, _i64p, _i64p]),
}

_lib = None


def load_library(path: Optional[str] = None):
  """Loads (once) and returns the ctypes library; raises if it has not been built."""
  global _lib
  if _lib is not None and path is None:
    return _lib
  # GC_LIB_VARIANT=<name>: an EXPERIMENT build of tools/build_variant.sh (csrc/variants/) -- timing-only ablations
  # that may compute WRONG values by construction.  Honoured only together with GC_ALLOW_EXPERIMENT_LIB=1, and loudly:
  # a stray variable in a real run must not swap the product library silently.
  variant = os.environ.get("GC_LIB_VARIANT")
  if variant and path is None:
    if os.environ.get("GC_ALLOW_EXPERIMENT_LIB") != "1":
      raise GencastHipError(
          f"GC_LIB_VARIANT={variant} asks for an experiment build (tools/build_variant.sh), which may compute wrong values; "
          "set GC_ALLOW_EXPERIMENT_LIB=1 as well to load it, or unset GC_LIB_VARIANT")
    print(f"[gencast_hip] WARNING: loading EXPERIMENT library variant {variant!r} (GC_LIB_VARIANT): timing only, "
          "results are not the product's", file=sys.stderr, flush=True)
  p = path or (os.path.join(_HERE, "csrc", "variants", f"libgencast_hip_{variant}.so") if variant else LIB_PATH)
  if not os.path.exists(p):
    raise GencastHipError(
        f"{p} not found: build it with gencast-flax-nnx_amd/csrc/build.sh "
        "(or __graft_entry__.build()); this package has no CPU fallback")
  lib = ctypes.CDLL(p)
  for name, (res, args) in SIGNATURES.items():
    fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
    fn.restype = res
    fn.argtypes = args
  if lib.gc_abi_version() != 1:
    raise GencastHipError("libgencast_hip.so ABI version mismatch")
  if path is None:
    _lib = lib
  return lib


def _f32(a) -> np.ndarray:
  return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a) -> np.ndarray:
  return np.ascontiguousarray(a, dtype=np.int32)


def _ptr(a: np.ndarray, ty):
  return a.ctypes.data_as(ty)


class NativeDenoiser:
  """Owns one `gc_handle` (one GPU, one stream).  Array-level API."""

  def __init__(self, *, latent_size, d_model, num_heads, ffw_hidden, num_layers, c_in, c_out,
               batch=1, device_id=0, noise_num_frequencies=32, noise_hidden=32,
               noise_base_period=16.0, hidden_layers=1):
    self._lib = load_library()
    self.cfg = GcConfig(latent_size, d_model, num_heads, ffw_hidden, num_layers, c_in, c_out, batch,
                        noise_num_frequencies, noise_hidden, noise_base_period)
    self._h = _hp()
    rc = self._lib.gc_create(ctypes.byref(self.cfg), device_id, ctypes.byref(self._h))
    if rc != GC_OK:
      msg = self._lib.gc_last_error(None).decode()
      self._h = None
      if rc in (GC_ERR_INVALID_ARGUMENT, GC_ERR_UNSUPPORTED):
        raise ValueError(msg)
      raise GencastHipError(f"gc_create failed ({rc}): {msg}")
    if int(hidden_layers) != 1:      # DenoiserArchitectureConfig.hidden_layers (denoiser.py:135): before any weight is loaded
      try:
        self.set_option("hidden_layers", str(int(hidden_layers)))
      except Exception:
        self.close()
        raise
    self.num_grid_nodes = None
    self.num_mesh_nodes = None

  # -- plumbing ------------------------------------------------------------------------------
  def _check(self, rc):
    if rc == GC_OK:
      return
    msg = self._lib.gc_last_error(self._h).decode()
    if rc in (GC_ERR_INVALID_ARGUMENT, GC_ERR_UNSUPPORTED):
      raise ValueError(msg)
    raise GencastHipError(f"libgencast_hip error {rc}: {msg}")

  def close(self):
    if getattr(self, "_h", None):
      self._lib.gc_destroy(self._h)
      self._h = None

  @property
  def closed(self) -> bool:
    return not getattr(self, "_h", None)

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass

  # -- set-up ----------------------------------------------------------------------------------
  def set_option(self, key: str, value: str) -> None:
    self._check(self._lib.gc_set_option(self._h, key.encode(), value.encode()))

  def set_graph(self, graph) -> None:
    """`graph`: geometry.DenoiserGraph (or anything with the same attributes)."""
    g = graph
    arrs = dict(
        g2m_s=_i32(g.g2m_senders), g2m_r=_i32(g.g2m_receivers), m2g_s=_i32(g.m2g_senders),
        m2g_r=_i32(g.m2g_receivers), rowptr=_i32(g.khop_rowptr), cols=_i32(g.khop_cols),
        gs=_f32(g.grid_struct), ms=_f32(g.mesh_struct), e1=_f32(g.g2m_edge_struct),
        e2=_f32(g.m2g_edge_struct))
    xyz = getattr(g, "mesh_xyz", None)
    xyz = None if xyz is None else _f32(xyz)
    G, M = int(g.num_grid_nodes), int(g.num_mesh_nodes)
    if arrs["gs"].shape != (G, 3) or arrs["ms"].shape != (M, 3):
      raise ValueError("structural node features must be [N,3]")
    if arrs["e1"].shape != (len(arrs["g2m_s"]), 4) or arrs["e2"].shape != (len(arrs["m2g_s"]), 4):
      raise ValueError("structural edge features must be [E,4]")
    if len(arrs["g2m_s"]) != len(arrs["g2m_r"]) or len(arrs["m2g_s"]) != len(arrs["m2g_r"]):
      raise ValueError("senders / receivers length mismatch")
    if arrs["rowptr"].shape != (M + 1,) or arrs["rowptr"][-1] != len(arrs["cols"]):
      raise ValueError("khop CSR is inconsistent")
    self._check(self._lib.gc_set_graph(
        self._h, G, M, len(arrs["g2m_s"]), _ptr(arrs["g2m_s"], _i32p), _ptr(arrs["g2m_r"], _i32p),
        len(arrs["m2g_s"]), _ptr(arrs["m2g_s"], _i32p), _ptr(arrs["m2g_r"], _i32p),
        _ptr(arrs["rowptr"], _i32p), _ptr(arrs["cols"], _i32p), _ptr(arrs["gs"], _f32p),
        _ptr(arrs["ms"], _f32p), _ptr(arrs["e1"], _f32p), _ptr(arrs["e2"], _f32p),
        None if xyz is None else _ptr(xyz, _f32p)))
    self.num_grid_nodes, self.num_mesh_nodes = G, M

  def load_weights(self, params: Dict[str, np.ndarray]) -> None:
    for name, w in params.items():
      a = _f32(w)
      shape = (ctypes.c_int64 * a.ndim)(*a.shape)
      self._check(self._lib.gc_load_weight(self._h, name.encode(), _ptr(a, _f32p), shape, a.ndim))

  def missing_weights(self) -> int:
    n = ctypes.c_int32(0)
    self._check(self._lib.gc_missing_weights(self._h, ctypes.byref(n)))
    return n.value

  def finalize(self) -> None:
    self._check(self._lib.gc_finalize(self._h))

  def set_noisy_slots(self, slots) -> None:
    s = _i32(slots)
    if s.shape != (self.cfg.c_out,):
      raise ValueError(f"expected {self.cfg.c_out} noisy slots")
    self._check(self._lib.gc_set_noisy_slots(self._h, _ptr(s, _i32p)))

  # -- compute -----------------------------------------------------------------------------------
  def _shape_in(self):
    return (self.num_grid_nodes, self.cfg.batch, self.cfg.c_in)

  def _shape_out(self):
    return (self.num_grid_nodes, self.cfg.batch, self.cfg.c_out)

  def denoise(self, grid_feats, sigma) -> np.ndarray:
    x = _f32(grid_feats)
    s = _f32(sigma).reshape(-1)
    if x.shape != self._shape_in():
      raise ValueError(f"grid_feats must be {self._shape_in()}, got {x.shape}")
    if s.shape != (self.cfg.batch,):
      raise ValueError("noise_levels expected to be shape (batch,).")
    out = np.empty(self._shape_out(), dtype=np.float32)
    self._check(self._lib.gc_denoise(self._h, _ptr(x, _f32p), _ptr(s, _f32p), _ptr(out, _f32p)))
    return out

  def sample(self, cond_feats, init_noise, sigmas, skip_dead_call=True):
    c, z, sg = _f32(cond_feats), _f32(init_noise), _f32(sigmas).reshape(-1)
    if c.shape != self._shape_in():
      raise ValueError(f"cond_feats must be {self._shape_in()}, got {c.shape}")
    if z.shape != self._shape_out():
      raise ValueError(f"init_noise must be {self._shape_out()}, got {z.shape}")
    out = np.empty(self._shape_out(), dtype=np.float32)
    st = GcSampleStats()
    self._check(self._lib.gc_sample(self._h, _ptr(c, _f32p), _ptr(z, _f32p), _ptr(sg, _f32p),
                                    len(sg) - 1, int(bool(skip_dead_call)), _ptr(out, _f32p),
                                    ctypes.byref(st)))
    return out, dict(denoiser_calls=st.denoiser_calls, device_ms=st.device_ms)

  def upload_cond(self, cond_feats):
    c = _f32(cond_feats)
    if c.shape != self._shape_in():
      raise ValueError(f"cond_feats must be {self._shape_in()}, got {c.shape}")
    self._check(self._lib.gc_upload_cond(self._h, _ptr(c, _f32p)))

  def upload_cond_dev(self, dev_ptr: int):
    self._check(self._lib.gc_upload_cond_dev(self._h, ctypes.c_void_p(dev_ptr)))

  def cond_device_ptr(self):
    p, n = ctypes.c_void_p(), ctypes.c_int64()
    self._check(self._lib.gc_cond_device_ptr(self._h, ctypes.byref(p), ctypes.byref(n)))
    return p.value, n.value

  def commit_cond(self):
    self._check(self._lib.gc_commit_cond(self._h))

  def upload_noise(self, init_noise):
    z = _f32(init_noise)
    if z.shape != self._shape_out():
      raise ValueError(f"init_noise must be {self._shape_out()}, got {z.shape}")
    self._check(self._lib.gc_upload_noise(self._h, _ptr(z, _f32p)))

  def sample_resident(self, sigmas, skip_dead_call=True, want_stats=True):
    sg = _f32(sigmas).reshape(-1)
    st = GcSampleStats()
    self._check(self._lib.gc_sample_resident(self._h, _ptr(sg, _f32p), len(sg) - 1,
                                             int(bool(skip_dead_call)),
                                             ctypes.byref(st) if want_stats else None))
    return dict(denoiser_calls=st.denoiser_calls, device_ms=st.device_ms) if want_stats else None

  def download_sample(self) -> np.ndarray:
    out = np.empty(self._shape_out(), dtype=np.float32)
    self._check(self._lib.gc_download_sample(self._h, _ptr(out, _f32p)))
    return out

  def stash_sample(self) -> None:
    """Waits for the last sample and snapshots it on the device (gc_stash_sample): the handle is free for the next step."""
    self._check(self._lib.gc_stash_sample(self._h))

  def download_stash(self) -> np.ndarray:
    """Copies the snapshot to the host on a side stream while the main stream keeps running (gc_download_stash)."""
    out = np.empty(self._shape_out(), dtype=np.float32)
    self._check(self._lib.gc_download_stash(self._h, _ptr(out, _f32p)))
    return out

  def rollout_plan(self, kind, src, sidx, a, b, n_forcing: int):
    """Per-channel context-update plan (gc_rollout_plan; kinds documented in gencast_hip.h)."""
    kind = np.ascontiguousarray(kind, dtype=np.int32)
    src = np.ascontiguousarray(src, dtype=np.int32)
    sidx = np.ascontiguousarray(sidx, dtype=np.int32)
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    for arr in (kind, src, sidx, a, b):
      if arr.shape != (self.cfg.c_in,):
        raise ValueError(f"plan arrays must have shape ({self.cfg.c_in},)")
    self._check(self._lib.gc_rollout_plan(self._h, _ptr(kind, _i32p), _ptr(src, _i32p), _ptr(sidx, _i32p),
                                          _ptr(a, _f32p), _ptr(b, _f32p), int(n_forcing)))
    self._n_forcing = int(n_forcing)

  def rollout_advance(self, forcings=None):
    """Applies the plan to the resident conditioning using the last sample (gc_rollout_advance)."""
    if forcings is None:
      self._check(self._lib.gc_rollout_advance(self._h, None))
      return
    forcings = np.ascontiguousarray(forcings, dtype=np.float32)
    want = (self.num_grid_nodes, self.cfg.batch, getattr(self, "_n_forcing", -1))
    if forcings.shape != want:
      raise ValueError(f"forcings must have shape {want}")
    self._check(self._lib.gc_rollout_advance(self._h, _ptr(forcings, _f32p)))

  def download_cond(self) -> np.ndarray:
    out = np.empty((self.num_grid_nodes, self.cfg.batch, self.cfg.c_in), dtype=np.float32)
    self._check(self._lib.gc_download_cond(self._h, _ptr(out, _f32p)))
    return out

  def sync(self):
    self._check(self._lib.gc_sync(self._h))

  # -- measurement ---------------------------------------------------------------------------------
  def kernel_classes(self):
    return [self._lib.gc_kernel_class_name(i).decode()
            for i in range(self._lib.gc_num_kernel_classes())]

  def profile_enable(self, cls: int):
    self._check(self._lib.gc_profile_enable(self._h, cls))

  def profile_set_stride(self, stride: int):
    self._check(self._lib.gc_profile_set_stride(self._h, stride))

  def profile_read(self):
    n, ms = ctypes.c_int32(), ctypes.c_float()
    self._check(self._lib.gc_profile_read(self._h, ctypes.byref(n), ctypes.byref(ms)))
    return n.value, ms.value

  def algorithmic_work(self):
    f, b = ctypes.c_double(), ctypes.c_double()
    self._check(self._lib.gc_algorithmic_work(self._h, ctypes.byref(f), ctypes.byref(b)))
    return f.value, b.value

  # -- spherical noise on the device, stochastic churn ----------------------------------------------
  def noise_set_tables(self, n_lat: int, n_lon: int, legendre, cos_table, sin_table) -> None:
    """Tables of `noise.SphericalNoise.device_tables()`: legendre [L, n_lat, L], cos / sin [n_lon, L]."""
    leg, ct, st = _f32(legendre), _f32(cos_table), _f32(sin_table)
    L = leg.shape[0]
    if leg.shape != (L, n_lat, L) or ct.shape != (n_lon, L) or st.shape != (n_lon, L):
      raise ValueError("noise tables must be legendre [L, n_lat, L], cos / sin [n_lon, L]")
    self._check(self._lib.gc_noise_set_tables(self._h, int(n_lat), int(n_lon), int(L), _ptr(leg, _f32p),
                                              _ptr(ct, _f32p), _ptr(st, _f32p)))

  def noise_seed(self, seed: int, stream: int = 0) -> None:
    self._check(self._lib.gc_noise_seed(self._h, ctypes.c_uint64(int(seed) & (2 ** 64 - 1)),
                                        ctypes.c_uint64(int(stream) & (2 ** 64 - 1))))

  def noise_draw(self) -> None:
    self._check(self._lib.gc_noise_draw(self._h))

  def download_noise(self) -> np.ndarray:
    out = np.empty(self._shape_out(), dtype=np.float32)
    self._check(self._lib.gc_download_noise(self._h, _ptr(out, _f32p)))
    return out

  def set_churn(self, rates, noise_level_inflation_factor: float = 1.0) -> None:
    """Per-step churn rates for the following sample calls (None / all zero: off)."""
    if rates is None:
      self._check(self._lib.gc_set_churn(self._h, None, 0, float(noise_level_inflation_factor)))
      return
    r = _f32(rates).reshape(-1)
    self._check(self._lib.gc_set_churn(self._h, _ptr(r, _f32p), len(r), float(noise_level_inflation_factor)))

  # -- ensemble exchange (RCCL inside the library) -------------------------------------------------
  def comm_init(self, unique_id: bytes, rank: int, world_size: int) -> None:
    if len(unique_id) != COMM_ID_BYTES:
      raise ValueError(f"unique id must be {COMM_ID_BYTES} bytes")
    buf = ctypes.create_string_buffer(bytes(unique_id), COMM_ID_BYTES)
    self._check(self._lib.gc_comm_init(self._h, buf, int(rank), int(world_size)))

  def comm_info(self):
    """(ranks, rank) as RCCL reports them for this handle's communicator; (0, -1) without one (gc_comm_info)."""
    n, r = ctypes.c_int32(), ctypes.c_int32()
    self._check(self._lib.gc_comm_info(self._h, ctypes.byref(n), ctypes.byref(r)))
    return n.value, r.value

  def comm_broadcast_cond(self, root: int = 0) -> None:
    self._check(self._lib.gc_comm_broadcast_cond(self._h, int(root)))

  def comm_allreduce_max(self, value: float) -> float:
    v = ctypes.c_double(float(value))
    self._check(self._lib.gc_comm_allreduce_max(self._h, ctypes.byref(v)))
    return v.value

  def comm_destroy(self) -> None:
    self._check(self._lib.gc_comm_destroy(self._h))

  def counter(self, name: str) -> int:
    """Named counters: range_fallbacks, launches_per_call, weights_f16_unsafe (gc_get_counter)."""
    v = ctypes.c_int64()
    self._check(self._lib.gc_get_counter(self._h, name.encode(), ctypes.byref(v)))
    return v.value

  # -- debug (tests) -------------------------------------------------------------------------------
  def debug_fetch(self, name: str) -> np.ndarray:
    r, c = ctypes.c_int64(), ctypes.c_int64()
    self._check(self._lib.gc_debug_fetch(self._h, name.encode(), None, 0, ctypes.byref(r),
                                         ctypes.byref(c)))
    out = np.empty((r.value, c.value), dtype=np.float32)
    self._check(self._lib.gc_debug_fetch(self._h, name.encode(), _ptr(out, _f32p), out.size,
                                         ctypes.byref(r), ctypes.byref(c)))
    return out

  def debug_set_layer_limit(self, n: int):
    self._check(self._lib.gc_debug_set_layer_limit(self._h, n))

  def debug_set_stop(self, layer: int, phase: int = 0):
    """Later forwards return inside block `layer` after phase 0 / 1 / 2 (gc_debug_set_stop); layer -1 = off."""
    self._check(self._lib.gc_debug_set_stop(self._h, int(layer), int(phase)))

  def debug_mesh_permutation(self) -> np.ndarray:
    p = np.empty(self.num_mesh_nodes, dtype=np.int32)
    self._check(self._lib.gc_debug_mesh_permutation(self._h, _ptr(p, _i32p)))
    return p

  def debug_attention_stats(self):
    a, b, c = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    self._check(self._lib.gc_debug_attention_stats(self._h, ctypes.byref(a), ctypes.byref(b),
                                                   ctypes.byref(c)))
    return dict(n_tiles=a.value, n_chunks=b.value, khop_nnz=c.value)


def comm_unique_id() -> bytes:
  """rank 0: the ncclUniqueId blob every rank passes to `NativeDenoiser.comm_init`."""
  lib = load_library()
  buf = ctypes.create_string_buffer(COMM_ID_BYTES)
  rc = lib.gc_comm_unique_id(buf)
  if rc != GC_OK:
    raise GencastHipError(f"gc_comm_unique_id failed ({rc}): {lib.gc_last_error(None).decode()}")
  return buf.raw


def device_count() -> int:
  return int(load_library().gc_device_count())


def device_pci_bus_id(device_id: int) -> str:
  """PCI bus id of a visible device: what the ranks of a launch compare to see whether two of them share a GPU."""
  buf = ctypes.create_string_buffer(64)
  rc = load_library().gc_device_pci_bus_id(int(device_id), buf, 64)
  if rc != 0:
    raise GencastHipError(f"gc_device_pci_bus_id({device_id}) failed with code {rc}")
  return buf.value.decode()
