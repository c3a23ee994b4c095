"""Channel stacking rules (common/model_utils.py:594-725; gencast/denoiser.py:770-830)."""
import numpy as np
import pytest

from gencast_flax_nnx_amd import config, datasets, synthetic
from gencast_flax_nnx_amd.datasets import Dataset, Variable
from gencast_flax_nnx_amd.denoiser import Denoiser


def test_variable_to_stacked_orders_and_broadcasts():
  b, t, lev, la, lo = 2, 2, 3, 4, 5
  a = np.arange(b * t * lev * la * lo, dtype=np.float32).reshape(b, t, lev, la, lo)
  v = Variable(("batch", "time", "level", "lat", "lon"), a)
  s = datasets.variable_to_stacked(v, v.sizes)
  assert s.shape == (b, la, lo, t * lev)
  # channel = t * n_level + l  (C order of the variable's own dims)
  np.testing.assert_array_equal(s[1, 2, 3, 1 * lev + 2], a[1, 1, 2, 2, 3])
  yp = Variable(("batch", "time"), np.array([[1.0, 2.0], [3.0, 4.0]], np.float32))
  s2 = datasets.variable_to_stacked(yp, dict(batch=b, lat=la, lon=lo))
  assert s2.shape == (b, la, lo, 2)
  np.testing.assert_array_equal(s2[1, :, :, 0], np.full((la, lo), 3.0))
  dp = Variable(("batch", "time", "lon"), np.arange(b * 1 * lo, dtype=np.float32).reshape(b, 1, lo))
  s3 = datasets.variable_to_stacked(dp, dict(batch=b, lat=la, lon=lo))
  np.testing.assert_array_equal(s3[1, 2, :, 0], dp.data[1, 0, :])


def test_dataset_to_stacked_sorted_names_and_roundtrip():
  rng = np.random.default_rng(0)
  ds = Dataset({
      "zeta": Variable(("batch", "time", "lat", "lon"), rng.standard_normal((1, 1, 3, 4)).astype(np.float32)),
      "alpha": Variable(("batch", "time", "level", "lat", "lon"), rng.standard_normal((1, 1, 2, 3, 4)).astype(np.float32)),
      "10m": Variable(("batch", "time", "lat", "lon"), rng.standard_normal((1, 1, 3, 4)).astype(np.float32)),
  }, coords=dict(lat=np.arange(3), lon=np.arange(4)))
  assert [n for n, _, _ in datasets.channel_layout(ds)] == ["10m", "alpha", "zeta"]   # ASCII order
  st = datasets.dataset_to_stacked(ds)
  assert st.shape == (1, 3, 4, 4)
  back = datasets.stacked_to_dataset(st, ds)
  for k in ds.keys():
    assert back[k].dims == ds[k].dims
    np.testing.assert_array_equal(back[k].data, ds[k].data)
  with pytest.raises(ValueError, match="Expected 4 channels but found 3"):
    datasets.stacked_to_dataset(st[..., :3], ds)


def test_stacked_to_dataset_requires_preserved_dims():
  ds = Dataset({"x": Variable(("batch", "time"), np.zeros((1, 1), np.float32))})
  with pytest.raises(ValueError, match="requires all Variables"):
    datasets.stacked_to_dataset(np.zeros((1, 2, 2, 1), np.float32), ds)


def test_nano_task_channel_accounting():
  """176 input + 86 forcing channels; noisy targets interleave by sorted name."""
  inp, tgt, frc = synthetic.make_example(batch=1)
  assert config.num_outputs(config.TASK) == 82
  feats, grid_shape, lat, lon, n_inputs = Denoiser.pack_inputs(inp, frc.assign(tgt))
  assert feats.shape == (10512, 1, 262) and grid_shape == (73, 144) and n_inputs == 176
  names = [n for n, _, _ in datasets.channel_layout(frc.assign(tgt))]
  assert names == ["10m_u_component_of_wind", "10m_v_component_of_wind", "2m_temperature",
                   "day_progress_cos", "day_progress_sin", "geopotential", "mean_sea_level_pressure",
                   "specific_humidity", "temperature", "u_component_of_wind", "v_component_of_wind",
                   "vertical_velocity", "year_progress_cos", "year_progress_sin"]
  d = Denoiser(None, config.nano_architecture())
  slots = d.noisy_slots(inp, frc, tgt)
  assert slots.shape == (82,) and len(set(slots.tolist())) == 82
  # writing the stacked targets into those slots reproduces pack_inputs of forcings ∪ targets
  cond, *_ = Denoiser.pack_inputs(inp, frc.assign(datasets.zeros_like(tgt)))
  tg = np.transpose(datasets.dataset_to_stacked(tgt), (1, 2, 0, 3)).reshape(10512, 1, 82)
  cond = cond.copy()
  cond[..., slots] = tg
  np.testing.assert_array_equal(cond, feats)
  # node index = lat_i * n_lon + lon_j
  t2m = inp["2m_temperature"].data
  name_off = {n: o for n, o, _ in datasets.channel_layout(inp)}
  np.testing.assert_array_equal(feats[5 * 144 + 7, 0, name_off["2m_temperature"]], t2m[0, 0, 5, 7])


def test_unpack_outputs_inverts_packing():
  inp, tgt, frc = synthetic.make_example(lat=np.linspace(-90, 90, 5), lon=np.arange(8) * 45.0, batch=2)
  st = np.transpose(datasets.dataset_to_stacked(tgt), (1, 2, 0, 3)).reshape(40, 2, 82)
  back = Denoiser.unpack_outputs(st, (5, 8), tgt)
  for k in tgt.keys():
    np.testing.assert_array_equal(back[k].data, tgt[k].data)
