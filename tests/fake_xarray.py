"""A minimal duck-typed stand-in for the `xarray` module -- TEST INFRASTRUCTURE ONLY (xarray is not installed in the
build container; the reference's harness is xarray throughout).  `tests/test_xarray_boundary.py` registers it as
`sys.modules["xarray"]` for the duration of a test; the package never imports this file.

Just what the boundary touches: `Dataset` (data_vars / coords / [] / keys / isel / sizes), `DataArray`
(values / dims / coords / name / ndim / shape / isel) and `concat(objs, dim)`.  Coordinates are 1-D, indexed by
their own dimension, as ERA5-style lat / lon / level / time / batch coordinates are."""
import numpy as np


class DataArray:
  def __init__(self, data, dims=None, coords=None, name=None):
    self.values = np.asarray(data)
    self.dims = tuple(dims) if dims is not None else tuple(f"dim_{i}" for i in range(self.values.ndim))
    if len(self.dims) != self.values.ndim:
      raise ValueError("dims do not match the data's rank")
    self.name = name
    self.coords = {}
    for k, v in dict(coords or {}).items():
      c = v if isinstance(v, DataArray) else DataArray(np.asarray(v), dims=(k,))
      if c.dims == (k,) and k in self.dims:
        if c.values.shape[0] != self.values.shape[self.dims.index(k)]:
          raise ValueError(f"coordinate {k!r} has length {c.values.shape[0]}, dimension has {self.values.shape[self.dims.index(k)]}")
        self.coords[k] = c

  ndim = property(lambda self: self.values.ndim)
  shape = property(lambda self: self.values.shape)
  sizes = property(lambda self: dict(zip(self.dims, self.values.shape)))

  def isel(self, **idx):
    sl = tuple(idx.get(d, slice(None)) for d in self.dims)
    coords = {k: DataArray(c.values[idx[k]] if k in idx else c.values, dims=(k,)) for k, c in self.coords.items()}
    return DataArray(self.values[sl], self.dims, coords, self.name)

  def __mul__(self, k):
    return DataArray(self.values * k, self.dims, self.coords, self.name)


class Dataset:
  def __init__(self, data_vars=None, coords=None):
    self.data_vars, self.coords = {}, {}
    for k, v in dict(coords or {}).items():
      self.coords[k] = v if isinstance(v, DataArray) else DataArray(np.asarray(v), dims=(k,))
    for k, v in dict(data_vars or {}).items():
      if not isinstance(v, DataArray):
        dims, data = v
        v = DataArray(data, dims, {d: self.coords[d] for d in dims if d in self.coords}, k)
      v.name = k
      self.data_vars[k] = v
      for ck, cv in v.coords.items():
        self.coords.setdefault(ck, cv)

  def __getitem__(self, k):
    return self.data_vars[k]

  def keys(self):
    return self.data_vars.keys()

  @property
  def sizes(self):
    out = {}
    for v in self.data_vars.values():
      out.update(v.sizes)
    return out

  def isel(self, **idx):
    return Dataset({k: v.isel(**{d: i for d, i in idx.items() if d in v.dims}) for k, v in self.data_vars.items()})

  def __mul__(self, k):
    return Dataset({n: v * k for n, v in self.data_vars.items()})


def concat(objs, dim):
  """xr.concat along an existing dimension (what the harness does with the per-step predictions)."""
  objs = list(objs)
  out = {}
  for name, first in objs[0].data_vars.items():
    ax = first.dims.index(dim)
    coords = dict(first.coords)
    if dim in coords:
      coords[dim] = DataArray(np.concatenate([o[name].coords[dim].values for o in objs]), dims=(dim,))
    out[name] = DataArray(np.concatenate([o[name].values for o in objs], axis=ax), first.dims, coords, name)
  return Dataset(out)
