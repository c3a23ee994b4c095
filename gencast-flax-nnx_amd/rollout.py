"""Autoregressive rollout around `GenCast.full_sampling`.

Three layers, each mirroring the reference at the array level:

* `normalize` / `unnormalize` / `InputsAndResiduals` -- common/normalization.py:31-238:
  inputs and forcings are normalised with per-variable (per-level) `scales`/`locations`;
  target variables that are also inputs are predicted as residuals to the last input frame,
  normalised with `residual_scales`; everything else with the plain statistics.
* `compose_next_frame` / `autoregressive_rollout` -- training/train_helpers.py:485-622
  (forecast-time branch): the context drops its oldest frame and gets a new one made of the
  predicted target variables, this step's forcings, and the input-only variables carried from
  the last context frame.
* `DeviceRollout` -- the same loop with the conditioning kept in HBM: in NORMALISED space the
  whole update (un-normalise, add last input, re-normalise, roll the time axis) is one affine
  per conditioning channel, which `gc_rollout_advance` applies on the device
  (include/gencast_hip.h).  Per step only the 4 progress forcings go up and the sample comes
  down; Python never re-packs the 11-MB conditioning.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np

from . import config as cfg
from . import datasets
from .datasets import Dataset, Variable
from .denoiser import Denoiser


# ---------------------------------------------------------------------------------------------
# small Dataset helpers (xarray's isel / concat on the time axis)
# ---------------------------------------------------------------------------------------------
def isel_time(ds: Dataset, sl) -> Dataset:
  """`ds.isel(time=sl)`: variables without a time axis pass through; an int drops the axis."""
  out: Dict[str, Variable] = {}
  for k, v in ds.items():
    if "time" not in v.dims:
      out[k] = v
      continue
    ax = v.dims.index("time")
    idx = [slice(None)] * len(v.dims)
    idx[ax] = sl
    data = v.data[tuple(idx)]
    dims = v.dims if isinstance(sl, slice) else tuple(d for d in v.dims if d != "time")
    out[k] = Variable(dims, data)
  coords = dict(ds.coords)
  if "time" in coords and coords["time"].ndim == 1:
    if isinstance(sl, slice):
      coords["time"] = coords["time"][sl]
    else:
      coords.pop("time")
  return Dataset(out, coords)


def concat_time(parts: Sequence[Dataset]) -> Dataset:
  """`xr.concat(parts, dim="time")` for datasets with the same variables."""
  first = parts[0]
  out: Dict[str, Variable] = {}
  for k, v in first.items():
    if "time" not in v.dims:
      out[k] = v
      continue
    ax = v.dims.index("time")
    out[k] = Variable(v.dims, np.concatenate([p[k].data for p in parts], axis=ax))
  coords = dict(first.coords)
  if all("time" in p.coords for p in parts):
    coords["time"] = np.concatenate([np.atleast_1d(p.coords["time"]) for p in parts])
  return Dataset(out, coords)


def _stat_like(stat: Variable, like: Variable) -> np.ndarray:
  """Broadcasts a statistic (dims a subset of `like`'s, e.g. () or ('level',)) to `like`."""
  shape = [1] * len(like.dims)
  for d, n in zip(stat.dims, np.shape(stat.data)):
    if d not in like.dims:
      raise ValueError(f"statistic dimension {d!r} is not a dimension of the variable {like.dims}")
    shape[like.dims.index(d)] = n
  order = sorted(range(len(stat.dims)), key=lambda i: like.dims.index(stat.dims[i]))
  data = np.transpose(np.asarray(stat.data), order) if stat.dims else np.asarray(stat.data)
  return data.reshape(shape).astype(like.data.dtype)


def normalize(values: Dataset, scales: Dataset, locations: Optional[Dataset]) -> Dataset:
  """common/normalization.py:31-50: (x - location) / scale, per variable, where statistics exist."""
  out = {}
  for name, v in values.items():
    data = v.data
    if locations is not None and name in locations:
      data = data - _stat_like(locations[name], v)
    if name in scales:
      data = data / _stat_like(scales[name], v)
    out[name] = Variable(v.dims, data)
  return Dataset(out, values.coords)


def unnormalize(values: Dataset, scales: Dataset, locations: Optional[Dataset]) -> Dataset:
  """common/normalization.py:53-71: x * scale + location."""
  out = {}
  for name, v in values.items():
    data = v.data
    if name in scales:
      data = data * _stat_like(scales[name], v)
    if locations is not None and name in locations:
      data = data + _stat_like(locations[name], v)
    out[name] = Variable(v.dims, data)
  return Dataset(out, values.coords)


class InputsAndResiduals:
  """common/normalization.py:74-238 (sampling side): normalised inputs, normalised residual targets."""

  def __init__(self, predictor, stddev_by_level: Dataset, mean_by_level: Dataset,
               diffs_stddev_by_level: Dataset):
    self.predictor = predictor
    self._scales = stddev_by_level
    self._locations = mean_by_level
    self._residual_scales = diffs_stddev_by_level
    self._residual_locations = None

  def _unnormalize_prediction_and_add_input(self, inputs: Dataset, name: str, pred: Variable) -> Variable:
    if pred.sizes.get("time") != 1:
      raise ValueError("normalization.InputsAndResiduals only supports predicting a single timestep.")
    one = Dataset({name: pred})
    if name in inputs:                                   # residual prediction (:110-118)
      out = unnormalize(one, self._residual_scales, self._residual_locations)[name]
      last = isel_time(Dataset({name: inputs[name]}), -1)[name]
      return Variable(out.dims, out.data + _broadcast_last(last, out))
    return unnormalize(one, self._scales, self._locations)[name]

  def _subtract_input_and_normalize_target(self, inputs: Dataset, name: str, target: Variable) -> Variable:
    if target.sizes.get("time") != 1:
      raise ValueError("normalization.InputsAndResiduals only supports wrapping predictors"
                       "that predict a single timestep.")
    if name in inputs:
      last = isel_time(Dataset({name: inputs[name]}), -1)[name]
      resid = Variable(target.dims, target.data - _broadcast_last(last, target))
      return normalize(Dataset({name: resid}), self._residual_scales, self._residual_locations)[name]
    return normalize(Dataset({name: target}), self._scales, self._locations)[name]

  def _wrap(self, fn_name, inputs, targets_template, forcings, **kwargs):
    given = (targets_template, inputs, forcings)            # xarray in -> xarray out (datasets.like_inputs)
    inputs = datasets.as_dataset(inputs)
    forcings = datasets.as_dataset(forcings)
    template = datasets.as_dataset(targets_template)
    norm_inputs = normalize(inputs, self._scales, self._locations)
    norm_forcings = normalize(forcings, self._scales, self._locations)
    norm_template = Dataset({k: self._subtract_input_and_normalize_target(inputs, k, v)
                             for k, v in template.items()}, template.coords)
    norm_pred = datasets.as_dataset(getattr(self.predictor, fn_name)(norm_inputs, norm_template, forcings=norm_forcings, **kwargs))
    return datasets.like_inputs(Dataset({k: self._unnormalize_prediction_and_add_input(inputs, k, v)
                                         for k, v in norm_pred.items()}, norm_pred.coords), *given)

  def full_sampling(self, inputs, targets_template, forcings, **kwargs):
    """common/normalization.py:200-238."""
    return self._wrap("full_sampling", inputs, targets_template, forcings, **kwargs)


def _broadcast_last(last: Variable, like: Variable) -> np.ndarray:
  """The last input frame (no time axis) broadcast against a time=1 variable."""
  shape = [1] * len(like.dims)
  for d, n in zip(last.dims, np.shape(last.data)):
    shape[like.dims.index(d)] = n
  order = sorted(range(len(last.dims)), key=lambda i: like.dims.index(last.dims[i]))
  return np.transpose(last.data, order).reshape(shape)


# ---------------------------------------------------------------------------------------------
# host-composed autoregressive rollout (training/train_helpers.py:485-622)
# ---------------------------------------------------------------------------------------------
def compose_next_frame(target_like: Dataset, forcings_like: Dataset, prev_context: Dataset,
                       task: cfg.TaskConfig = cfg.TASK) -> Dataset:
  """One full input frame (time=1): predicted targets + this step's forcings + carried input-only vars."""
  target_vars, forcing_vars = set(task.target_variables), set(task.forcing_variables)
  input_only = set(task.input_variables) - target_vars - forcing_vars
  dv: Dict[str, Variable] = {}
  for v in target_vars:
    if v in target_like:
      dv[v] = target_like[v]
  for v in forcing_vars:
    if v in forcings_like:
      dv[v] = forcings_like[v]
  for v in input_only:
    if v in prev_context:
      var = prev_context[v]
      if "time" in var.dims:
        ax = var.dims.index("time")
        dv[v] = Variable(var.dims, np.take(var.data, [-1], axis=ax))
      else:
        dv[v] = var
  coords = dict(prev_context.coords)
  coords.update(target_like.coords)
  return Dataset(dv, coords)


def autoregressive_rollout(model, inputs: Dataset, targets: Dataset, forcings: Dataset, horizon: int, *,
                           context_steps: int = 2, task: cfg.TaskConfig = cfg.TASK,
                           init_noise: Optional[Sequence[np.ndarray]] = None):
  """Forecast-time AR rollout.  Returns (mse vs `targets`, predictions [time=horizon], targets[:horizon]).

  `targets` supplies the per-step templates (`* 0`) and the MSE reference exactly as the reference
  does (train_helpers.py:596-640); `init_noise[k]` optionally fixes step k's initial noise.
  """
  given = (targets, inputs, forcings)
  inputs, targets, forcings = (datasets.as_dataset(x) for x in (inputs, targets, forcings))
  if inputs.sizes.get("time", 0) < context_steps:
    raise ValueError(f"inputs carry {inputs.sizes.get('time', 0)} time steps (need at least {context_steps})")
  context = isel_time(inputs, slice(-context_steps, None))
  preds: List[Dataset] = []
  for k in range(horizon):
    template = isel_time(targets, slice(k, k + 1)).map(np.zeros_like)
    forc_k = isel_time(forcings, slice(k, k + 1))
    kw = {} if init_noise is None else {"init_noise": init_noise[k]}
    pred = datasets.as_dataset(model.full_sampling(inputs=context, targets_template=template, forcings=forc_k, **kw))
    preds.append(pred)
    tail = isel_time(context, slice(1, None))
    frame = compose_next_frame(pred, forc_k, context, task)
    context = concat_time([tail, Dataset({n: frame[n] for n in tail.keys()}, frame.coords)])
  rollout = concat_time(preds)
  future = isel_time(targets, slice(0, horizon))
  se, cnt = 0.0, 0
  for name, v in rollout.items():
    d = v.data.astype(np.float64) - future[name].data.astype(np.float64)
    se += float((d * d).sum())
    cnt += d.size
  future_x = isel_time(datasets.as_dataset(given[0]), slice(0, horizon))
  if datasets.is_xarray(given[0]):                          # predictions on the targets' time axis, like the harness's xr.concat
    return se / max(cnt, 1), datasets.to_xarray(rollout, given[0].isel(time=slice(0, horizon))), given[0].isel(time=slice(0, horizon))
  return se / max(cnt, 1), datasets.like_inputs(rollout, None, *given[1:]), future_x


# ---------------------------------------------------------------------------------------------
# device-resident rollout
# ---------------------------------------------------------------------------------------------
def _time_blocks(ds: Dataset):
  """For every variable in stacking order: (name, first_channel, n_time, channels_per_time)."""
  out = []
  for name, off, n in datasets.channel_layout(ds):
    v = ds[name]
    rest = [d for d in v.dims if d not in datasets.PRESERVED]
    if "time" in rest:
      if rest[0] != "time":
        raise ValueError(f"{name}: the time axis must come before {rest[1:]} for the device rollout")
      nt = v.sizes["time"]
    else:
      nt = 1
    out.append((name, off, nt, n // nt))
  return out


def _per_channel_stat(stat: Optional[Dataset], name: str, var: Variable, default: float) -> np.ndarray:
  """Statistic of one time slice of `var`, flattened in channel order (level-major remainder)."""
  rest = [d for d in var.dims if d not in datasets.PRESERVED and d != "time"]
  n = int(np.prod([var.sizes[d] for d in rest])) if rest else 1
  if stat is None or name not in stat:
    return np.full(n, default, np.float64)
  s = stat[name]
  shape = [var.sizes[d] for d in rest]
  full = np.ones(shape, np.float64)
  view = [1] * len(rest)
  for d, m in zip(s.dims, np.shape(s.data)):
    view[rest.index(d)] = m
  order = sorted(range(len(s.dims)), key=lambda i: rest.index(s.dims[i]))
  data = np.transpose(np.asarray(s.data, np.float64), order) if s.dims else np.asarray(s.data, np.float64)
  return (full * data.reshape(view)).reshape(-1)


def build_rollout_plan(inputs: Dataset, forcings: Dataset, template: Dataset, task: cfg.TaskConfig,
                       norm: Optional[InputsAndResiduals]):
  """Plan arrays for `gc_rollout_plan` + the order of the forcing columns `advance` expects.

  In normalised space, with s/l the input statistics and rs/rl the residual ones (normalization.py:100-121):
    x_new = pred * rs + rl + x_last,  xn = (x - l) / s   =>   xn_new = xn_last + (rs / s) * pred + rl / s.
  Without a normalisation wrapper the prediction is the new frame itself (a = 1, no carry).
  """
  n_inputs = sum(n for _, _, n in datasets.channel_layout(inputs))
  merged = forcings.assign(datasets.zeros_like(template))
  c_in = n_inputs + sum(n for _, _, n in datasets.channel_layout(merged))
  kind = np.zeros(c_in, np.int32)
  src = np.zeros(c_in, np.int32)
  sidx = np.zeros(c_in, np.int32)
  a = np.zeros(c_in, np.float32)
  b = np.zeros(c_in, np.float32)
  tgt_off = {name: off for name, off, _ in datasets.channel_layout(template)}
  forc_off = {name: n_inputs + off for name, off, _ in datasets.channel_layout(merged)}
  target_vars, forcing_vars = set(task.target_variables), set(task.forcing_variables)

  for name, off, nt, per in _time_blocks(inputs):
    if nt < 2:
      continue                                            # no time axis (or one frame): carried unchanged
    for t in range(nt - 1):                               # older slots take the next-newer slot's values
      c = off + t * per + np.arange(per)
      kind[c] = 1
      src[c] = c + per
    last = off + (nt - 1) * per + np.arange(per)
    if name in target_vars and name in template:
      var = inputs[name]
      if norm is not None:
        s = _per_channel_stat(norm._scales, name, var, 1.0)
        rs = _per_channel_stat(norm._residual_scales, name, var, 1.0)
        rl = _per_channel_stat(norm._residual_locations, name, var, 0.0)
        kind[last] = 2
        src[last] = last
        a[last] = (rs / s).astype(np.float32)
        b[last] = (rl / s).astype(np.float32)
      else:
        kind[last] = 4
        a[last] = 1.0
      sidx[last] = tgt_off[name] + np.arange(per)
    elif name in forcing_vars and name in forcings:
      kind[last] = 1                                      # this step's forcing becomes the newest input frame
      src[last] = forc_off[name] + np.arange(per)
    # input-only variables: newest slot keeps its value (carried forward)

  forcing_cols = []                                       # (name, channel offset in the NEXT forcings array)
  nf = 0
  for name, off, n in datasets.channel_layout(merged):
    c = n_inputs + off + np.arange(n)
    if name in forcings and name not in template:
      kind[c] = 3
      sidx[c] = nf + np.arange(n)
      forcing_cols.append((name, nf, n))
      nf += n
    # noisy-target slots are rewritten by the sampler at every denoiser call: left as they are
  return dict(kind=kind, src=src, sidx=sidx, a=a, b=b, n_forcing=nf), forcing_cols


class DeviceRollout:
  """AR rollout with the conditioning resident in HBM (one `gc_rollout_advance` per step)."""

  def __init__(self, model, norm: Optional[InputsAndResiduals] = None, task: cfg.TaskConfig = cfg.TASK,
               device_noise: bool = False):
    """`device_noise`: draw every step's initial state on the GPU (`gc_noise_draw`) instead of synthesising
    it on the host and uploading 3.4 MB per step."""
    self.model = model                                    # a GenCast (its sampler drives the native handle)
    self.norm = norm
    self.task = task
    self.device_noise = device_noise
    self.last_step_ms: List[float] = []

  def _forcing_rows(self, forc_k: Dataset, forcing_cols, sizes, grid_shape) -> np.ndarray:
    if self.norm is not None:
      forc_k = normalize(forc_k, self.norm._scales, self.norm._locations)
    cols = []
    for name, _, _ in forcing_cols:
      cols.append(datasets.variable_to_stacked(forc_k[name], sizes))
    st = np.concatenate(cols, axis=-1)                    # (batch, lat, lon, nf)
    a = np.transpose(st, (1, 2, 0, 3))
    return np.ascontiguousarray(a.reshape((grid_shape[0] * grid_shape[1],) + a.shape[2:]), np.float32)

  def run(self, inputs: Dataset, targets: Dataset, forcings: Dataset, horizon: int, *,
          context_steps: int = 2, init_noise: Optional[Sequence[np.ndarray]] = None, rngs=0):
    """Returns the predictions (physical units, time = horizon) like `autoregressive_rollout`."""
    import time as _time
    given = (targets, inputs, forcings)
    inputs, targets, forcings = (datasets.as_dataset(x) for x in (inputs, targets, forcings))
    context = isel_time(inputs, slice(-context_steps, None))
    template0 = isel_time(targets, slice(0, 1)).map(np.zeros_like)
    forc0 = isel_time(forcings, slice(0, 1))
    sampler = self.model._sampler
    den: Denoiser = self.model.denoiser
    if self.norm is not None:
      n_in = normalize(context, self.norm._scales, self.norm._locations)
      n_fo = normalize(forc0, self.norm._scales, self.norm._locations)
    else:
      n_in, n_fo = context, forc0
    cond, grid_shape, slots = den.init_for(n_in, template0, n_fo)
    native = den.native
    native.set_noisy_slots(slots)
    plan, forcing_cols = build_rollout_plan(context, forc0, template0, self.task, self.norm)
    native.rollout_plan(**plan)
    native.upload_cond(cond)
    sizes = dict(forc0.sizes)
    sizes.update(context.sizes)
    gen = rngs if isinstance(rngs, np.random.Generator) else np.random.default_rng(rngs)
    sigmas = np.asarray(sampler.noise_levels, np.float32)
    shape = (cond.shape[0], cond.shape[1], den.dims.c_out)
    last_phys = {k: isel_time(Dataset({k: context[k]}), -1)[k] for k in template0.keys() if k in context}
    preds: List[Dataset] = []
    self.last_step_ms = []

    def finish(k, out):
      """Host post-processing of step k's sample: unpack, un-normalise, add the last physical frame."""
      template = isel_time(targets, slice(k, k + 1)).map(np.zeros_like)
      norm_pred = Denoiser.unpack_outputs(out, grid_shape, template)
      if self.norm is not None:
        dv = {}
        for name, v in norm_pred.items():
          one = Dataset({name: v})
          if name in last_phys:
            u = unnormalize(one, self.norm._residual_scales, self.norm._residual_locations)[name]
            dv[name] = Variable(u.dims, u.data + _broadcast_last(last_phys[name], u))
          else:
            dv[name] = unnormalize(one, self.norm._scales, self.norm._locations)[name]
        pred = Dataset(dv, norm_pred.coords)
      else:
        pred = norm_pred
      for name in last_phys:
        last_phys[name] = isel_time(Dataset({name: pred[name]}), -1)[name]
      preds.append(pred)

    on_device = self.device_noise and init_noise is None
    churn = getattr(sampler, "_stochastic_churn", False)
    if on_device or churn:
      sampler.ensure_device_noise(native, template0)
      native.noise_seed(sampler.seed_from(gen), 0)
    native.set_churn(sampler._per_step_churn_rates if churn else None,
                     getattr(sampler, "_noise_level_inflation_factor", 1.0))

    def draw(k):
      if on_device:
        return None
      return np.asarray(init_noise[k], np.float32) if init_noise is not None else \
          sampler.draw_noise(gen, shape, template0)

    noise = draw(0)
    pending = None                                        # step whose snapshot waits to be downloaded and finished
    for k in range(horizon):
      t0 = _time.perf_counter()
      if on_device:
        native.noise_draw()
      else:
        native.upload_noise(noise)
      native.sample_resident(sigmas, skip_dead_call=True, want_stats=False)   # asynchronous
      # while the GPU samples step k: download step k-1's snapshot (side stream), finish it on the host, draw the next noise
      if pending is not None:
        finish(pending, native.download_stash())
      if k + 1 < horizon:
        noise = draw(k + 1)
        frows = (self._forcing_rows(isel_time(forcings, slice(k + 1, k + 2)), forcing_cols, sizes, grid_shape)
                 if plan["n_forcing"] else None)
      native.stash_sample()                               # waits for sample k (and its domain check); device-side snapshot
      if k + 1 < horizon:
        native.rollout_advance(frows)                     # the gap to the next sample holds no host copy any more
      pending = k
      self.last_step_ms.append(1e3 * (_time.perf_counter() - t0))
    finish(pending, native.download_stash())
    out = concat_time(preds)
    if datasets.is_xarray(given[0]):                        # predictions on the targets' time axis, like the harness's xr.concat
      return datasets.to_xarray(out, given[0].isel(time=slice(0, horizon)))
    return datasets.like_inputs(out, None, *given[1:])
