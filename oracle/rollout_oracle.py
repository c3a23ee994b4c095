"""CPU restatement of the autoregressive rollout step (TEST INFRASTRUCTURE ONLY).

Only tests/ may import this.  Plain dicts {name: (dims, ndarray)} stand in for xarray Datasets.
Follows, read as text:
  * common/normalization.py:31-71 (normalize / unnormalize), :100-140 (residual targets),
    :200-238 (`full_sampling`);
  * training/train_helpers.py:485-547 (`_compose_next_frame`), :596-622 (forecast-time loop).
Parity status: unpinned -- the reference holds no fixtures for the rollout and cannot be
imported here (xarray / jax absent); this restatement is cross-checked against the product's
host implementation and against the packed-array plan the device path executes.
"""
import numpy as np


def _bcast(stat, dims_like, shape_like, dtype):
  sdims, sdata = stat
  shape = [1] * len(dims_like)
  for d, n in zip(sdims, np.shape(sdata)):
    shape[dims_like.index(d)] = n
  order = sorted(range(len(sdims)), key=lambda i: dims_like.index(sdims[i]))
  data = np.transpose(np.asarray(sdata), order) if sdims else np.asarray(sdata)
  return data.reshape(shape).astype(dtype)


def normalize(values, scales, locations):
  """normalization.py:31-50."""
  out = {}
  for name, (dims, data) in values.items():
    if locations is not None and name in locations:
      data = data - _bcast(locations[name], dims, data.shape, data.dtype)
    if name in scales:
      data = data / _bcast(scales[name], dims, data.shape, data.dtype)
    out[name] = (dims, data)
  return out


def unnormalize(values, scales, locations):
  """normalization.py:53-71."""
  out = {}
  for name, (dims, data) in values.items():
    if name in scales:
      data = data * _bcast(scales[name], dims, data.shape, data.dtype)
    if locations is not None and name in locations:
      data = data + _bcast(locations[name], dims, data.shape, data.dtype)
    out[name] = (dims, data)
  return out


def isel_time(ds, sl):
  out = {}
  for name, (dims, data) in ds.items():
    if "time" not in dims:
      out[name] = (dims, data)
      continue
    idx = [slice(None)] * len(dims)
    idx[dims.index("time")] = sl
    nd = dims if isinstance(sl, slice) else tuple(d for d in dims if d != "time")
    out[name] = (nd, data[tuple(idx)])
  return out


def _last_like(last, dims_like):
  ldims, ldata = last
  shape = [1] * len(dims_like)
  for d, n in zip(ldims, ldata.shape):
    shape[dims_like.index(d)] = n
  order = sorted(range(len(ldims)), key=lambda i: dims_like.index(ldims[i]))
  return np.transpose(ldata, order).reshape(shape)


def full_sampling_normalized(inner, inputs, template, forcings, stats, **kw):
  """normalization.py:200-238.  stats = (scales, locations, residual_scales); residual locations None."""
  scales, locations, rscales = stats
  n_in = normalize(inputs, scales, locations)
  n_fo = normalize(forcings, scales, locations)
  n_tpl = {}
  for name, (dims, data) in template.items():
    if name in inputs:                                            # :131-137
      last = isel_time({name: inputs[name]}, -1)[name]
      n_tpl[name] = normalize({name: (dims, data - _last_like(last, dims))}, rscales, None)[name]
    else:
      n_tpl[name] = normalize({name: (dims, data)}, scales, locations)[name]
  n_pred = inner(n_in, n_tpl, n_fo, **kw)
  out = {}
  for name, (dims, data) in n_pred.items():
    if name in inputs:                                            # :106-118
      u = unnormalize({name: (dims, data)}, rscales, None)[name][1]
      last = isel_time({name: inputs[name]}, -1)[name]
      out[name] = (dims, u + _last_like(last, dims))
    else:
      out[name] = unnormalize({name: (dims, data)}, scales, locations)[name]
  return out


def compose_next_frame(target_like, forcings_like, prev_context, task):
  """train_helpers.py:485-547."""
  tv, fv = set(task.target_variables), set(task.forcing_variables)
  io = set(task.input_variables) - tv - fv
  out = {}
  for v in tv:
    if v in target_like:
      out[v] = target_like[v]
  for v in fv:
    if v in forcings_like:
      out[v] = forcings_like[v]
  for v in io:
    if v in prev_context:
      dims, data = prev_context[v]
      if "time" in dims:
        out[v] = (dims, np.take(data, [-1], axis=dims.index("time")))
      else:
        out[v] = (dims, data)
  return out


def concat_time(parts):
  out = {}
  for name, (dims, data) in parts[0].items():
    if "time" not in dims:
      out[name] = (dims, data)
    else:
      out[name] = (dims, np.concatenate([p[name][1] for p in parts], axis=dims.index("time")))
  return out


def autoregressive_rollout(sample_fn, inputs, targets, forcings, horizon, task, context_steps=2):
  """train_helpers.py:596-622.  sample_fn(context, template, forcings_k, k) -> prediction dict."""
  context = isel_time(inputs, slice(-context_steps, None))
  preds = []
  for k in range(horizon):
    template = {n: (d, np.zeros_like(a)) for n, (d, a) in isel_time(targets, slice(k, k + 1)).items()}
    forc_k = isel_time(forcings, slice(k, k + 1))
    pred = sample_fn(context, template, forc_k, k)
    preds.append(pred)
    tail = isel_time(context, slice(1, None))
    frame = compose_next_frame(pred, forc_k, context, task)
    context = concat_time([tail, {n: frame[n] for n in tail}])
  return concat_time(preds), context


def apply_plan(cond, sample, forcings, plan):
  """Array-level statement of gc_rollout_advance (include/gencast_hip.h)."""
  new = np.empty_like(cond)
  for c in range(cond.shape[-1]):
    k = plan["kind"][c]
    if k == 0:
      new[..., c] = cond[..., c]
    elif k == 1:
      new[..., c] = cond[..., plan["src"][c]]
    elif k == 2:
      new[..., c] = cond[..., plan["src"][c]] + plan["a"][c] * sample[..., plan["sidx"][c]] + plan["b"][c]
    elif k == 3:
      new[..., c] = forcings[..., plan["sidx"][c]]
    else:
      new[..., c] = plan["a"][c] * sample[..., plan["sidx"][c]] + plan["b"][c]
  return new
