"""Host-side mirror of the reference's call contracts: construction, argument
errors, schedules (no GPU needed)."""
import numpy as np
import pytest

from gencast_flax_nnx_amd import (Denoiser, GenCast, Sampler, config, create_gencast_model, datasets,
                                  noise_schedule, stochastic_churn_rate_schedule, synthetic, weights)
from oracle import gencast_oracle as O


def test_noise_schedule_is_the_references():
  np.testing.assert_allclose(noise_schedule(80.0, 0.03, 20, 7.0), O.noise_schedule(80.0, 0.03, 20, 7.0), rtol=0)
  lv = noise_schedule()
  assert len(lv) == 31 and lv[0] == 80.0 and lv[-1] == 0.0 and np.all(np.diff(lv) < 0)
  np.testing.assert_allclose(lv[-2], 0.002)
  np.testing.assert_array_equal(stochastic_churn_rate_schedule(lv, 0.0), np.zeros(30))


def test_param_specs_counts():
  """nano 23.06 M live parameters (the dead m2g mesh update excluded)."""
  d = weights.ModelDims(c_in=262, c_out=82, latent=256, d_model=256, num_heads=4, ffw_hidden=2048, num_layers=16)
  n = weights.count_params(d)
  dead = 2 * (256 * 256 + 256) + 16 * 512 + 512
  assert abs((n + dead) / 1e6 - 23.06) < 0.02
  p = weights.random_params(d, seed=3)
  assert set(p) == set(weights.param_specs(d))
  assert all(v.dtype == np.float32 for v in p.values())
  k = p["denoiser.predictor.mesh_gnn.batch_first_transformer.blocks.0.attn_module.final_linear.kernel"]
  assert 0.5 < k.std() * 16 < 1.5      # O(1/sqrt(fan_in)), not the reference's degenerate zero init


def test_model_dims_validation():
  with pytest.raises(ValueError, match="num_heads"):
    weights.ModelDims(c_in=10, c_out=2, latent=128, d_model=128, num_heads=3, ffw_hidden=128, num_layers=1)
  with pytest.raises(ValueError, match="latent"):
    weights.ModelDims(c_in=10, c_out=2, latent=128, d_model=256, num_heads=2, ffw_hidden=128, num_layers=1)


def test_denoiser_argument_errors():
  arch = config.nano_architecture()
  d = Denoiser(None, arch)
  inp, tgt, frc = synthetic.make_example(lat=np.linspace(-90, 90, 5), lon=np.arange(8) * 45.0)
  with pytest.raises(ValueError, match=r"noise_levels expected to be shape \(batch,\)"):
    d(inp, tgt, np.ones((1, 1), np.float32), frc)
  with pytest.raises(ValueError, match=r"noise_levels expected to be shape \(batch,\)"):
    d(inp, tgt, np.ones((3,), np.float32), frc)
  with pytest.raises(ValueError, match="node_output_size"):
    d(inp, tgt, np.ones((1,), np.float32), frc)
  import dataclasses
  bad = dataclasses.replace(arch, sparse_transformer_config=dataclasses.replace(
      arch.sparse_transformer_config, attention_type="splash_mha"))
  with pytest.raises(NotImplementedError):
    Denoiser(None, bad)
  for n in (0, 5):                                           # DenoiserArchitectureConfig.hidden_layers (denoiser.py:135)
    with pytest.raises(ValueError, match="hidden_layers"):
      Denoiser(None, dataclasses.replace(arch, hidden_layers=n))
  d2 = Denoiser(None, dataclasses.replace(arch, hidden_layers=2))     # accepted (runs as a chain of launches per MLP)
  from gencast_flax_nnx_amd import weights
  from gencast_flax_nnx_amd.denoiser import dims_from_arch
  one, two = (weights.param_specs(dims_from_arch(dataclasses.replace(arch, hidden_layers=n), 262, 82)) for n in (1, 2))
  extra = sorted(set(two) - set(one))
  assert len(extra) == 20 and all(".network.network.layers.4." in k for k in extra) and not set(one) - set(two)
  L = arch.latent_size
  dec = f"{weights.P_M2G}.decoder_network.embed_node_fns.grid_nodes.network.network.layers."
  assert two[dec + "2.kernel"] == (L, L) and two[dec + "4.kernel"] == (L, 82) and one[dec + "2.kernel"] == (L, 82)
  del d2


def test_sampler_config_and_churn():
  gc = create_gencast_model(mesh_size=2, d_model=128, num_layers=1, num_heads=2)
  assert isinstance(gc, GenCast)
  assert gc.denoiser._arch.node_output_size == 82
  s = gc._sampler
  np.testing.assert_allclose(s.noise_levels, O.noise_schedule(80.0, 0.03, 20, 7.0))
  sc = config.SamplerConfig()
  assert (sc.stochastic_churn_rate, sc.churn_min_noise_level, sc.noise_level_inflation_factor) == (2.5, 0.75, 1.05)
  import dataclasses
  churny = Sampler(gc.denoiser, **dataclasses.asdict(sc))
  assert churny._stochastic_churn
  rates = churny._per_step_churn_rates                       # samplers_utils.py:415-431
  np.testing.assert_allclose(rates, O.stochastic_churn_rate_schedule(churny.noise_levels, 2.5, sc.churn_min_noise_level, sc.churn_max_noise_level))
  assert rates.max() == pytest.approx(2.5 / 20) and (rates > 0).sum() == ((churny.noise_levels[:-1] >= sc.churn_min_noise_level) & (churny.noise_levels[:-1] <= sc.churn_max_noise_level)).sum()
  assert Sampler.seed_from(7) == 7 and Sampler.seed_from(np.random.default_rng(1)) == Sampler.seed_from(np.random.default_rng(1))
  with pytest.raises(ValueError, match="rngs"):
    Sampler.seed_from(None)
  inp, tgt, frc = synthetic.make_example(lat=np.linspace(-90, 90, 5), lon=np.arange(8) * 45.0)
  with pytest.raises(NotImplementedError):
    gc.loss(inp, tgt, frc)


def test_nan_cleaner_fills_and_reintroduces():
  """gencast/nan_cleaning.py:27-156 on a variable with NaN 'land' points."""
  from gencast_flax_nnx_amd import NaNCleaner
  from gencast_flax_nnx_amd.datasets import Dataset, Variable
  rng = np.random.default_rng(0)
  sst = rng.standard_normal((1, 2, 3, 4)).astype(np.float32)
  sst[0, :, 1, 2] = np.nan                                  # same land point in both input frames
  sst[0, 1, 0, 0] = np.nan                                  # NaN in one frame only
  other = rng.standard_normal((1, 2, 3, 4)).astype(np.float32)
  dims = ("batch", "time", "lat", "lon")
  inputs = Dataset({"sst": Variable(dims, sst), "t2m": Variable(dims, other)}, coords=dict(lat=np.arange(3), lon=np.arange(4)))
  fill = Dataset({"sst": Variable((), np.float32(-7.0))})
  seen = {}

  class Inner:
    def __call__(self, inp, tmpl, frc, **kw):
      seen["call"] = inp
      return Dataset({"sst": Variable(dims, np.ones((1, 1, 3, 4), np.float32)), "t2m": Variable(dims, np.ones((1, 1, 3, 4), np.float32))})

    def full_sampling(self, inp, tmpl, frc, **kw):
      seen["fs"] = (inp, frc, kw)
      return self(inp, tmpl, frc)
  for reintro in (False, True):
    nc = NaNCleaner(Inner(), "sst", fill, reintroduce_nans=reintro)
    out = nc.full_sampling(inputs, None, inputs, tag=1)
    got, frc, kw = seen["fs"]
    assert kw == {"tag": 1}
    assert not np.isnan(got["sst"].data).any() and not np.isnan(frc["sst"].data).any()
    assert got["sst"].data[0, 0, 1, 2] == -7.0 and got["sst"].data[0, 1, 0, 0] == -7.0
    np.testing.assert_array_equal(got["sst"].data[0, 0, 0, 1:], sst[0, 0, 0, 1:])       # untouched elsewhere
    np.testing.assert_array_equal(got["t2m"].data, other)
    assert np.isnan(inputs["sst"].data).sum() == 3                                       # caller's data unchanged
    nanmask = np.isnan(out["sst"].data[0, 0])
    if reintro:
      want = np.zeros((3, 4), bool)
      want[1, 2] = want[0, 0] = True                                                     # ANY frame NaN (:65)
      np.testing.assert_array_equal(nanmask, want)
      assert not np.isnan(out["t2m"].data).any()
    else:
      assert not nanmask.any()
    nc(inputs, None, None)
    assert not np.isnan(seen["call"]["sst"].data).any()
  # TASK has no such variable: pass-through
  nc = NaNCleaner(Inner(), "sea_surface_temperature", Dataset({"sea_surface_temperature": Variable((), np.float32(0))}))
  nc.full_sampling(inputs, None, None)
  assert seen["fs"][0] is inputs or np.isnan(seen["fs"][0]["sst"].data).sum() == 3
  with pytest.raises(NotImplementedError):
    nc.loss()


def test_predictor_fn_adapter_and_graph_injection_arguments():
  """GenCast.as_predictor_fn has the reference's PredictorFn signature (common/rollout.py:29-38); Denoiser
  accepts an injected graph / options without touching the GPU."""
  import inspect
  from gencast_flax_nnx_amd import geometry
  lat, lon = np.linspace(-90, 90, 5), np.arange(8) * 45.0
  gr = geometry.build_denoiser_graph(grid_lat=lat, grid_lon=lon, mesh_size=1, attention_k_hop=1)
  gc = GenCast(config.TASK, config.nano_architecture(mesh_size=1), config.SamplerConfig(stochastic_churn_rate=0.0),
               graph=gr, options={"precision": "f32"})
  assert gc.denoiser._graph_arg is gr and gc.denoiser._options == {"precision": "f32"}
  fn = gc.as_predictor_fn()
  assert list(inspect.signature(fn).parameters)[:4] == ["rng", "inputs", "targets_template", "forcings"]
  calls = {}
  gc._sampler = lambda inputs, tmpl, frc, rngs=None, **kw: calls.update(dict(a=(inputs, tmpl, frc, rngs, kw))) or "pred"
  assert fn(5, "i", "t", "f", extra=1) == "pred" and calls["a"] == ("i", "t", "f", 5, {"extra": 1})


def test_reference_checkpoint_state_import():
  """`clean_state` (training/evaluation.py:137-176) + flattening on a synthetic nested
  state that mirrors the NNX paths of 8a-W and carries everything the reference's clean-up removes:
  normalisation datasets, private buffers, {0: leaf} singletons, one-element lists, the `graph_network`
  wrapper (hoisted), `.value` leaves and the dead mesh2grid mesh update."""
  d = weights.ModelDims(c_in=20, c_out=6, latent=128, d_model=128, num_heads=2, ffw_hidden=256, num_layers=2)
  params = weights.random_params(d, seed=1)
  nested = {}
  for name, arr in params.items():
    parts = name.split(".")
    cur = nested
    for p in parts[:-1]:
      cur = cur.setdefault(int(p) if p.isdigit() and p != "0" else p, {})
    style = hash(name) % 3
    cur[parts[-1]] = {"value": arr} if style == 0 else ({0: arr} if style == 1 else [arr])
  nested["stddev_by_level"] = {"2m_temperature": np.ones(3), "10m_u_component_of_wind": np.ones(3)}
  nested["denoiser"]["_private_buffer"] = np.zeros(4)
  nested["denoiser"]["predictor"]["mesh2grid_gnn"]["processor_networks"]["0"]["graph_network"].setdefault(
      "update_node_fns", {}).setdefault("mesh_nodes", {"node_fn": {"dead": {"value": np.zeros(2)}}})
  nested["optimizer_state"] = {"mu": [np.zeros(3)]}
  got = weights.import_reference_state(nested, d)
  assert sorted(got) == sorted(params)
  for k in params:
    np.testing.assert_array_equal(got[k], params[k])
  cleaned = weights.clean_state(nested)
  assert "stddev_by_level" not in cleaned and "_private_buffer" not in cleaned["denoiser"]
  flat = weights.flatten_state(cleaned)
  assert not any(".graph_network." in n for n in flat)        # the wrapper was hoisted, as in the reference
  bad = dict(nested)
  bad["denoiser"] = {k: v for k, v in nested["denoiser"].items() if k != "noise_level_encoder"}
  with pytest.raises(ValueError, match="missing"):
    weights.import_reference_state(bad, d)
  assert len(weights.import_reference_state(bad, d, strict=False)) == len(params) - 4


def test_reference_constructor_surface_without_a_gpu():
  """gencast/gencast.py:145-154 and gencast/denoiser.py:153-159: `gpu_mesh` is accepted (and ignored), `rngs` may
  be an nnx.Rngs-like object; `grid2mesh_aggregate_normalization` becomes a library option instead of raising."""
  import dataclasses

  class Rngs:
    def noise(self):
      return np.array([1, 2], np.uint32)

    def params(self):
      return np.array([3, 4], np.uint32)

  r = Rngs()
  arch = dataclasses.replace(config.nano_architecture(mesh_size=1), grid2mesh_aggregate_normalization=2.0)
  gc = GenCast(config.TASK, arch, config.SamplerConfig(stochastic_churn_rate=0.0), None, None, "a-jax-mesh", r)
  assert gc.rngs is r
  assert gc.denoiser._options["grid2mesh_aggregate_normalization"] == "2.0"
  assert GenCast(config.TASK, config.nano_architecture(mesh_size=1), rngs=7).rngs.integers(0, 10) == np.random.default_rng(7).integers(0, 10)
  den = Denoiser(None, arch, None, rngs=r, gpu_mesh=object())
  assert den._param_seed == 7 and "grid2mesh_aggregate_normalization" in den._options
  with pytest.raises(ValueError):
    Denoiser(None, dataclasses.replace(arch, grid2mesh_aggregate_normalization=-1.0))


def test_typed_prng_keys_from_an_nnx_rngs_stream_are_unwrapped():
  """Current flax hands out TYPED jax keys from `rngs.params()` / `rngs.noise()`; `np.asarray` on one raises TypeError
  (ADVICE r4).  `datasets.key_words` unwraps them (jax.random.key_data when jax imports; here a stub that exposes what a
  typed key exposes: no __array__, a `_base_array`), and an opaque object falls back to a hash instead of raising."""
  from gencast_flax_nnx_amd import datasets, sampler

  class TypedKey:                       # like jax's PRNGKeyArray: np.asarray(key) -> TypeError
    def __init__(self, words):
      self._base_array = np.asarray(words, np.uint32)

    def __array__(self, *a, **k):
      raise TypeError("JAX array with PRNGKey dtype cannot be converted to a NumPy array")

  class Rngs:
    def noise(self):
      return TypedKey([1, 2])

    def params(self):
      return TypedKey([3, 4])

  np.testing.assert_array_equal(datasets.key_words(TypedKey([1, 2])), [1, 2])
  np.testing.assert_array_equal(datasets.key_words(np.array([5, 6], np.uint32)), [5, 6])      # legacy raw keys
  opaque = datasets.key_words(object)                                                       # nothing to unwrap: hashed
  assert opaque.dtype == np.uint32 and opaque.size == 2 and np.array_equal(opaque, datasets.key_words(object))
  r = Rngs()
  den = Denoiser(None, config.nano_architecture(mesh_size=1), None, rngs=r)
  assert den._param_seed == 7
  a = sampler._draw_noise(r, (4, 1, 3))
  b = np.random.default_rng([1, 2]).standard_normal((4, 1, 3), dtype=np.float32)
  np.testing.assert_array_equal(a, b)
  assert sampler.Sampler.seed_from(r) == sampler.Sampler.seed_from(type("R", (), {"noise": lambda self: np.array([1, 2], np.uint32)})())
