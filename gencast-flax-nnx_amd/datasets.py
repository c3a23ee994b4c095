"""Minimal labelled-array containers + the reference's channel stacking rules.

xarray is not available where this runs, so `Dataset` / `Variable` here carry
just what the path needs (dims, data, lat/lon coords).  The stacking functions
restate common/model_utils.py:594-725 (`variable_to_stacked`,
`dataset_to_stacked`, `stacked_to_dataset`) and :145-167 (lat/lon leading axes):
variables in SORTED-name order; every dim other than (batch, lat, lon) is folded
into channels in the variable's own dim order (C order); missing preserved dims
are broadcast.  `from_xarray` / `to_xarray` convert when xarray is importable.
"""
from __future__ import annotations

import dataclasses
from typing import Dict, Mapping, Optional, Sequence, Tuple

import numpy as np

PRESERVED = ("batch", "lat", "lon")


@dataclasses.dataclass
class Variable:
  dims: Tuple[str, ...]
  data: np.ndarray

  def __post_init__(self):
    self.dims = tuple(self.dims)
    if len(self.dims) != np.ndim(self.data):
      raise ValueError(f"dims {self.dims} do not match data of rank {np.ndim(self.data)}")

  @property
  def sizes(self) -> Dict[str, int]:
    return dict(zip(self.dims, np.shape(self.data)))


class Dataset:
  """name -> Variable, plus coordinate arrays (at least lat / lon when present)."""

  def __init__(self, data_vars: Optional[Mapping[str, Variable]] = None,
               coords: Optional[Mapping[str, np.ndarray]] = None):
    self.data_vars: Dict[str, Variable] = {}
    for k, v in (data_vars or {}).items():
      self.data_vars[k] = v if isinstance(v, Variable) else Variable(*v)
    self.coords: Dict[str, np.ndarray] = {k: np.asarray(v) for k, v in (coords or {}).items()}

  def __getitem__(self, k) -> Variable:
    return self.data_vars[k]

  def __contains__(self, k) -> bool:
    return k in self.data_vars

  def keys(self):
    return self.data_vars.keys()

  def items(self):
    return self.data_vars.items()

  @property
  def sizes(self) -> Dict[str, int]:
    out: Dict[str, int] = {}
    for v in self.data_vars.values():
      for d, n in v.sizes.items():
        if out.setdefault(d, n) != n:
          raise ValueError(f"conflicting sizes for dimension {d!r}")
    for d, c in self.coords.items():
      if c.ndim == 1:
        out.setdefault(d, c.shape[0])
    return out

  def assign(self, other: "Dataset") -> "Dataset":
    """xarray's Dataset.assign(other): other's variables replace / extend ours."""
    dv = dict(self.data_vars)
    dv.update(other.data_vars)
    co = dict(self.coords)
    co.update(other.coords)
    return Dataset(dv, co)

  def map(self, fn) -> "Dataset":
    return Dataset({k: Variable(v.dims, fn(v.data)) for k, v in self.items()}, self.coords)


def variable_to_stacked(var: Variable, sizes: Mapping[str, int],
                        preserved_dims: Sequence[str] = PRESERVED) -> np.ndarray:
  """-> array with axes preserved_dims + (channels,) (model_utils.py:594-626)."""
  stack_dims = [d for d in var.dims if d not in preserved_dims]
  keep = [d for d in preserved_dims if d in var.dims]
  perm = [var.dims.index(d) for d in keep] + [var.dims.index(d) for d in stack_dims]
  a = np.transpose(np.asarray(var.data), perm)
  a = a.reshape(a.shape[:len(keep)] + (-1,))               # channels = C-order of stack_dims
  # insert the missing preserved dims and broadcast them
  shape, src = [], 0
  idx = []
  for d in preserved_dims:
    if d in keep:
      shape.append(a.shape[src])
      idx.append(slice(None))
      src += 1
    else:
      shape.append(int(sizes[d]))
      idx.append(None)
  a = a[tuple(idx) + (slice(None),)]
  return np.broadcast_to(a, tuple(shape) + (a.shape[-1],))


def dataset_to_stacked(ds: Dataset, sizes: Optional[Mapping[str, int]] = None,
                       preserved_dims: Sequence[str] = PRESERVED) -> np.ndarray:
  """model_utils.py:629-659: concat of sorted variables along channels."""
  sizes = sizes or ds.sizes
  parts = [variable_to_stacked(ds[name], sizes, preserved_dims) for name in sorted(ds.keys())]
  if not parts:
    shape = tuple(int(sizes[d]) for d in preserved_dims)
    return np.zeros(shape + (0,), dtype=np.float32)
  return np.concatenate(parts, axis=-1)


def channel_layout(ds: Dataset, preserved_dims: Sequence[str] = PRESERVED):
  """[(name, first_channel, n_channels)] in stacking order."""
  out, off = [], 0
  for name in sorted(ds.keys()):
    n = 1
    for d, s in ds[name].sizes.items():
      if d not in preserved_dims:
        n *= s
    out.append((name, off, n))
    off += n
  return out


def stacked_to_dataset(stacked: np.ndarray, template: Dataset,
                       preserved_dims: Sequence[str] = PRESERVED) -> Dataset:
  """Inverse of dataset_to_stacked for a template (model_utils.py:662-725).

  `stacked` has axes preserved_dims + (channels,).
  """
  names = sorted(template.keys())
  total = 0
  for name in names:
    tv = template[name]
    if not all(d in tv.dims for d in preserved_dims):
      raise ValueError(f"stacked_to_dataset requires all Variables to have {tuple(preserved_dims)} "
                       f"dimensions, but found only {tv.dims}.")
  layout = channel_layout(template, preserved_dims)
  total = sum(n for _, _, n in layout)
  if total != stacked.shape[-1]:
    raise ValueError(f"Expected {total} channels but found {stacked.shape[-1]}, when trying to "
                     f"convert a stacked array of shape {stacked.shape} to a dataset.")
  out = {}
  for name, off, n in layout:
    tv = template[name]
    un_dims = [d for d in tv.dims if d not in preserved_dims]
    un_sizes = [tv.sizes[d] for d in un_dims]
    a = stacked[..., off:off + n].reshape(stacked.shape[:-1] + tuple(un_sizes))
    cur = list(preserved_dims) + un_dims
    a = np.transpose(a, [cur.index(d) for d in tv.dims])
    out[name] = Variable(tv.dims, np.ascontiguousarray(a))
  return Dataset(out, template.coords)


def zeros_like(ds: Dataset) -> Dataset:
  return ds.map(lambda a: np.zeros_like(np.asarray(a, dtype=np.float32)))


# ---- optional xarray interop -------------------------------------------------------------------

def from_xarray(xds) -> Dataset:
  coords = {k: np.asarray(v.values) for k, v in xds.coords.items() if v.ndim == 1}
  return Dataset({k: Variable(tuple(v.dims), np.asarray(v.values)) for k, v in xds.data_vars.items()},
                 coords)


def to_xarray(ds: Dataset, template=None):
  """-> xarray.Dataset.  With an xarray `template`: every variable takes the template variable's coordinates and name
  (as `xarray.DataArray(data=..., dims=..., coords=template[name].coords)` in model_utils.py:716-723); variables the
  template lacks, and the no-template case, take this Dataset's own 1-D coordinates."""
  import xarray  # pylint: disable=import-outside-toplevel
  own = {k: v for k, v in ds.coords.items() if v.ndim == 1}
  out = {}
  for k in ds.keys():
    v = ds[k]
    if template is not None and k in template.data_vars:
      tv = template[k]
      out[k] = xarray.DataArray(v.data, dims=v.dims, coords=tv.coords, name=getattr(tv, "name", k))
    else:
      out[k] = xarray.DataArray(v.data, dims=v.dims, coords={d: own[d] for d in v.dims if d in own}, name=k)
  return xarray.Dataset(out)


def is_xarray(obj) -> bool:
  """An xarray.Dataset-like argument (duck-typed: `data_vars` + `coords`) as opposed to this package's `Dataset`."""
  return obj is not None and not isinstance(obj, Dataset) and hasattr(obj, "data_vars") and hasattr(obj, "coords")


def like_inputs(out: Dataset, template, *others):
  """What a call at the reference boundary returns.  The reference's harness hands xarray Datasets to `Denoiser` /
  `Sampler` / `full_sampling` and goes on with xarray operations on the result
  (`xr.concat(preds_per_step, dim="time")`, training/train_helpers.py:569-586,607-624), so when ANY of the
  arguments arrived as an xarray object the result is an xarray.Dataset carrying the template's dims, coordinates
  and variable names (gencast/denoiser.py:809-830 builds its result from the template the same way); callers that
  work with this package's `Dataset` throughout get a `Dataset` back."""
  if not any(is_xarray(o) for o in (template,) + others):
    return out
  return to_xarray(out, template if is_xarray(template) else None)


def as_dataset(obj) -> Dataset:
  if isinstance(obj, Dataset):
    return obj
  if obj is None:
    return Dataset()
  if hasattr(obj, "data_vars") and hasattr(obj, "coords"):
    return from_xarray(obj)
  raise TypeError(f"cannot interpret {type(obj)} as a Dataset")


def key_words(key) -> np.ndarray:
  """uint32 words of a random key as an `nnx.Rngs` stream hands it out (`rngs.params()`, `rngs.noise()`;
  gencast/gencast.py:289-294, gencast/denoiser.py:153-159).  Current flax returns a TYPED jax key, on which
  `np.asarray` raises TypeError: it is unwrapped with `jax.random.key_data` when jax is importable (or through the key's
  own `_base_array` / `__array__`-free duck interface), legacy raw `uint32[2]` keys and plain integer sequences pass as
  they are, and anything else is hashed by its repr -- deterministic for one object state, never an exception at the
  port's call site."""
  try:
    return np.asarray(key).astype(np.uint32).ravel()
  except (TypeError, ValueError):
    pass
  try:                                               # typed PRNG key (jax >= 0.4.16)
    import jax  # noqa: PLC0415  (absent in this image: the branch is covered by a stub in tests/test_host_api.py)
    return np.asarray(jax.random.key_data(key)).astype(np.uint32).ravel()
  except Exception:  # noqa: BLE001
    pass
  for attr in ("_base_array", "key_data"):
    raw = getattr(key, attr, None)
    raw = raw() if callable(raw) else raw
    if raw is not None:
      try:
        return np.asarray(raw).astype(np.uint32).ravel()
      except (TypeError, ValueError):
        pass
  import hashlib  # noqa: PLC0415
  return np.frombuffer(hashlib.sha256(repr(key).encode()).digest()[:8], dtype=np.uint32).copy()

