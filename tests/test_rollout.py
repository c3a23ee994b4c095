"""Autoregressive rollout: host logic vs the CPU restatement, and the
per-channel plan the device path executes vs re-packing the host-composed context."""
import numpy as np
import pytest

from gencast_flax_nnx_amd import config as cfg
from gencast_flax_nnx_amd import datasets, rollout, synthetic
from gencast_flax_nnx_amd.datasets import Dataset, Variable
from gencast_flax_nnx_amd.denoiser import Denoiser
from oracle import rollout_oracle as RO


def _as_dict(ds):
  return {k: (v.dims, v.data) for k, v in ds.items()}


def _stats(task, seed=5):
  rng = np.random.default_rng(seed)
  nlev = len(task.pressure_levels)
  def mk(lo, hi, center=0.0):
    out = {}
    for name in set(task.input_variables) | set(task.target_variables):
      if name in cfg.ALL_ATMOSPHERIC_VARS:
        out[name] = Variable(("level",), (center + rng.uniform(lo, hi, nlev)).astype(np.float32))
      else:
        out[name] = Variable((), np.float32(center + rng.uniform(lo, hi)))
    return Dataset(out)
  return mk(0.5, 2.0), mk(-1.0, 1.0), mk(0.1, 0.5)


def _example(horizon, batch=2, seed=0):
  lat = np.linspace(-90, 90, 7)
  lon = np.arange(0, 360, 30.0)
  task = cfg.TASK
  inputs, tgt1, forc1 = synthetic.make_example(lat, lon, batch=batch, seed=seed)
  rng = np.random.default_rng(seed + 1)
  def stretch(ds, nt):
    out = {}
    for k, v in ds.items():
      ax = v.dims.index("time")
      shape = list(v.data.shape)
      shape[ax] = nt
      out[k] = Variable(v.dims, rng.standard_normal(shape).astype(np.float32))
    return Dataset(out, ds.coords)
  return task, inputs, stretch(tgt1, horizon), stretch(forc1, horizon)


class FakePredictor:
  """Deterministic stand-in for GenCast.full_sampling in normalised space."""

  def full_sampling(self, inputs, targets_template, forcings=None, init_noise=None, **kw):
    out = {}
    f = sum(float(np.mean(v.data)) for v in forcings.data_vars.values())
    for name, v in targets_template.items():
      last = rollout.isel_time(Dataset({name: inputs[name]}), -1)[name]
      base = rollout._broadcast_last(last, v)
      out[name] = Variable(v.dims, (0.3 * np.tanh(base) + 0.01 * f + 0 * v.data).astype(np.float32))
    return Dataset(out, targets_template.coords)


def test_normalize_roundtrip_and_broadcast():
  task, inputs, targets, forcings = _example(1)
  scales, locs, _ = _stats(task)
  n = rollout.normalize(inputs, scales, locs)
  back = rollout.unnormalize(n, scales, locs)
  for k, v in inputs.items():
    np.testing.assert_allclose(back[k].data, v.data, rtol=2e-6, atol=2e-6)
  ref = RO.normalize(_as_dict(inputs), _as_dict(scales), _as_dict(locs))
  for k in ref:
    np.testing.assert_array_equal(n[k].data, ref[k][1])


def test_host_rollout_matches_oracle():
  horizon = 3
  task, inputs, targets, forcings = _example(horizon)
  stats = _stats(task)
  model = rollout.InputsAndResiduals(FakePredictor(), *stats)
  mse, preds, future = rollout.autoregressive_rollout(model, inputs, targets, forcings, horizon, task=task)
  fake = FakePredictor()
  sd = tuple(_as_dict(s) for s in (stats[0], stats[1], stats[2]))

  def inner(n_in, n_tpl, n_fo, **kw):
    to_ds = lambda d: Dataset({k: Variable(*v) for k, v in d.items()})
    return _as_dict(fake.full_sampling(to_ds(n_in), to_ds(n_tpl), to_ds(n_fo)))

  def sample_fn(context, template, forc_k, k):
    return RO.full_sampling_normalized(inner, context, template, forc_k, sd)

  ref, _ = RO.autoregressive_rollout(sample_fn, _as_dict(inputs), _as_dict(targets), _as_dict(forcings),
                                     horizon, task)
  for name, (dims, data) in ref.items():
    assert preds[name].dims == dims
    np.testing.assert_allclose(preds[name].data, data, rtol=1e-6, atol=1e-6)
  assert preds[next(iter(ref))].sizes["time"] == horizon
  assert np.isfinite(mse) and mse > 0


@pytest.mark.parametrize("with_norm", [True, False])
def test_device_plan_equals_repacking_the_composed_context(with_norm):
  """apply_plan(packed cond, sample) == pack(next context, next forcings) on every channel the
  sampler does not overwrite -- the identity gc_rollout_advance relies on."""
  horizon = 2
  task, inputs, targets, forcings = _example(horizon, batch=2, seed=3)
  stats = _stats(task)
  norm = rollout.InputsAndResiduals(FakePredictor(), *stats) if with_norm else None
  context = rollout.isel_time(inputs, slice(-2, None))
  template = rollout.isel_time(targets, slice(0, 1)).map(np.zeros_like)
  forc0 = rollout.isel_time(forcings, slice(0, 1))
  forc1 = rollout.isel_time(forcings, slice(1, 2))
  plan, forcing_cols = rollout.build_rollout_plan(context, forc0, template, task, norm)

  def pack(ctx, fo):
    if norm is not None:
      ctx = rollout.normalize(ctx, stats[0], stats[1])
      fo = rollout.normalize(fo, stats[0], stats[1])
    feats, grid_shape, _, _, _ = Denoiser.pack_inputs(ctx, fo.assign(datasets.zeros_like(template)))
    return feats, grid_shape

  cond0, grid_shape = pack(context, forc0)
  rng = np.random.default_rng(9)
  c_out = sum(n for _, _, n in datasets.channel_layout(template))
  sample = rng.standard_normal((cond0.shape[0], cond0.shape[1], c_out)).astype(np.float32)
  norm_pred = Denoiser.unpack_outputs(sample, grid_shape, template)
  if norm is not None:
    pred = Dataset({k: norm._unnormalize_prediction_and_add_input(context, k, v) for k, v in norm_pred.items()},
                   norm_pred.coords)
  else:
    pred = norm_pred
  tail = rollout.isel_time(context, slice(1, None))
  frame = rollout.compose_next_frame(pred, forc0, context, task)
  nxt = rollout.concat_time([tail, Dataset({n: frame[n] for n in tail.keys()}, frame.coords)])
  expect, _ = pack(nxt, forc1)

  dr = rollout.DeviceRollout(model=None, norm=norm, task=task)
  sizes = dict(forc0.sizes)
  sizes.update(context.sizes)
  frows = dr._forcing_rows(forc1, forcing_cols, sizes, grid_shape)
  got = RO.apply_plan(cond0, sample, frows, plan)
  n_inputs = sum(n for _, _, n in datasets.channel_layout(context))
  slots = set()
  merged = forc0.assign(datasets.zeros_like(template))
  for name, off, n in datasets.channel_layout(merged):
    if name in template:
      slots.update(range(n_inputs + off, n_inputs + off + n))
  keep = np.array([c for c in range(cond0.shape[-1]) if c not in slots])
  np.testing.assert_allclose(got[..., keep], expect[..., keep], rtol=2e-5, atol=2e-5)
  assert set(np.unique(plan["kind"])) <= {0, 1, 2, 3, 4}
  assert plan["n_forcing"] == sum(n for _, _, n in forcing_cols) == 4
