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
  inp, tgt, frc = synthetic.make_example(lat=np.linspace(-90, 90, 5), lon=np.arange(8) * 45.0)
  with pytest.raises(NotImplementedError, match="churn"):
    churny(inp, tgt, frc, rngs=0)
  with pytest.raises(NotImplementedError):
    gc.loss(inp, tgt, frc)
