"""The oracle checked against what little the reference pins, against independent
implementations of the third-party primitives (torch), and by internal identities."""
import os

import numpy as np
import pytest
import torch

from oracle import gencast_oracle as O
from tests import helpers

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "denoiser_tiny.npz"))


def test_noise_schedule_known_values():
  """gencast/samplers_utils.py:395-412 evaluated for the nano sampler config."""
  want = [80, 62.081269, 47.718984, 36.304321, 27.314867, 20.305066, 14.897415, 10.774344,
          7.670781, 5.367349, 3.684189, 2.475358, 1.623786, 1.036763, 0.641921, 0.38368,
          0.220146, 0.120405, 0.062206, 0.03, 0]
  np.testing.assert_allclose(O.noise_schedule(80.0, 0.03, 20, 7.0), want, atol=5e-7)


def test_churn_schedule():
  lv = O.noise_schedule(80.0, 0.03, 20, 7.0)
  assert not O.stochastic_churn_rate_schedule(lv, 0.0).any()
  r = O.stochastic_churn_rate_schedule(lv, 2.5, 0.75, float("inf"))
  assert r.shape == (20,)
  np.testing.assert_allclose(r[lv[:-1] >= 0.75], 2.5 / 20)
  assert not r[lv[:-1] < 0.75].any()
  # clamp at sqrt(2)-1
  np.testing.assert_allclose(O.stochastic_churn_rate_schedule(lv, 1000.0, 0.0, 100.0).max(), np.sqrt(2) - 1)


def test_primitives_against_torch():
  rng = np.random.default_rng(0)
  x = rng.standard_normal((7, 3, 64)) * 3 + 0.5
  t = torch.from_numpy(x)
  np.testing.assert_allclose(O.layer_norm(x), torch.nn.functional.layer_norm(t, (64,), eps=1e-6).numpy(), atol=1e-12)
  np.testing.assert_allclose(O.gelu_tanh(x), torch.nn.functional.gelu(t, approximate="tanh").numpy(), atol=1e-12)
  np.testing.assert_allclose(O.swish(x), torch.nn.functional.silu(t).numpy(), atol=1e-12)
  np.testing.assert_allclose(O._softmax_lastaxis(x), torch.softmax(t, -1).numpy(), atol=1e-12)
  w, b = rng.standard_normal((64, 5)), rng.standard_normal(5)
  np.testing.assert_allclose(O.linear(x, w, b), torch.nn.functional.linear(t, torch.from_numpy(w.T), torch.from_numpy(b)).numpy(), atol=1e-12)


def test_mlp_with_several_hidden_layers_against_torch():
  """common/mlp.py:157-199: `mlp_num_hidden_layers` x (Linear, activation) + the output Linear, an nnx.Sequential whose
  i-th Linear is `layers.{2 i}`; checked against a torch.nn.Sequential with the same weights (1, 2 and 3 hidden layers)."""
  rng = np.random.default_rng(4)
  x = rng.standard_normal((9, 2, 12))
  for n in (1, 2, 3):
    params, mods, width = {}, [], 12
    for i in range(n + 1):
      out = 5 if i == n else 16
      w, b = rng.standard_normal((width, out)) / np.sqrt(width), 0.1 * rng.standard_normal(out)
      params[f"m.network.network.layers.{2 * i}.kernel"], params[f"m.network.network.layers.{2 * i}.bias"] = w, b
      lin = torch.nn.Linear(width, out).double()
      with torch.no_grad():
        lin.weight.copy_(torch.from_numpy(w.T)); lin.bias.copy_(torch.from_numpy(b))
      mods += [lin] + ([torch.nn.SiLU()] if i < n else [])
      width = out
    with torch.no_grad():
      ref = torch.nn.Sequential(*mods)(torch.from_numpy(x)).numpy()
    np.testing.assert_allclose(O.mlp(params, "m", x, O.swish), ref, atol=1e-12)


def test_fourier_features_layout():
  v = np.array([0.0, 1.0])
  f = O.fourier_features(v, 16.0, 32)
  assert f.shape == (2, 64)
  np.testing.assert_allclose(f[0], [1] * 32 + [0] * 32, atol=1e-15)
  k = np.arange(1, 33)
  np.testing.assert_allclose(f[1, :32], np.cos(2 * np.pi * k / 16))
  np.testing.assert_allclose(f[1, 32:], np.sin(2 * np.pi * k / 16))


def test_segment_sum_is_incidence_matmul():
  rng = np.random.default_rng(1)
  data = rng.standard_normal((50, 2, 8))
  ids = rng.integers(0, 7, size=50)
  inc = np.zeros((7, 50))
  inc[ids, np.arange(50)] = 1
  np.testing.assert_allclose(O.segment_sum(data, ids, 7), np.einsum("se,ebc->sbc", inc, data), atol=1e-12)


def test_attention_formulations_agree():
  gr, dims, params, x, sigma = helpers.tiny_setup(batch=2)
  gd = helpers.graph_dict(gr)
  rng = np.random.default_rng(2)
  m = gr.num_mesh_nodes
  q, k, v = (rng.standard_normal((2, m, 2, 16)) for _ in range(3))
  outs = [O.make_attention_fn(gd, f)(q, k, v) for f in ("neighbour", "dense", "triblock")]
  np.testing.assert_allclose(outs[0], outs[1], atol=1e-12)
  np.testing.assert_allclose(outs[0], outs[2], atol=1e-12)
  # rows of the attention sum to one over exactly the neighbourhood: constant v is reproduced
  ones = np.ones_like(v)
  np.testing.assert_allclose(O.make_attention_fn(gd, "triblock")(q, k, ones), ones, atol=1e-12)


def test_split_w1_edge_mlp_equals_concat():
  """typed_graph_net.py:303: concat([e, s, r]) @ W1 == e@Wa + s@Wb + r@Wc."""
  rng = np.random.default_rng(3)
  e, s, r = (rng.standard_normal((9, 1, 4)) for _ in range(3))
  w = rng.standard_normal((12, 5))
  cat = O.linear(np.concatenate([e, s, r], -1), w)
  np.testing.assert_allclose(cat, O.linear(e, w[:4]) + O.linear(s, w[4:8]) + O.linear(r, w[8:]), atol=1e-12)


def test_denoiser_forward_matches_golden_and_f32():
  gr, dims, params, x, sigma = helpers.tiny_setup(batch=2)
  assert np.isclose(x.astype(np.float64).sum(), GOLD["x_sum"])
  gd = helpers.graph_dict(gr)
  y = O.denoiser_forward(params, gd, x, sigma, num_layers=dims.num_layers, num_heads=dims.num_heads,
                         attention="triblock")
  np.testing.assert_allclose(y, GOLD["y"], atol=1e-11)
  y32 = O.denoiser_forward(params, gd, x, sigma, num_layers=dims.num_layers, num_heads=dims.num_heads,
                           attention="dense", dtype=np.float32)
  assert y32.dtype == np.float32
  np.testing.assert_allclose(y32, GOLD["y"], atol=1e-4)


def test_mesh_numbering_is_internal():
  """Permuting mesh nodes (with the edge lists) leaves grid outputs unchanged."""
  gr, dims, params, x, sigma = helpers.tiny_setup(batch=1)
  gd = helpers.graph_dict(gr)
  m = gr.num_mesh_nodes
  perm = np.random.default_rng(4).permutation(m)       # new -> old
  inv = np.argsort(perm)
  import scipy.sparse
  mask = scipy.sparse.csr_matrix((np.ones(len(gr.khop_cols), np.int8), gr.khop_cols, gr.khop_rowptr), shape=(m, m))
  pm = mask[perm][:, perm].tocsr()
  pm.sort_indices()
  gp = dict(gd, g2m_receivers=inv[gr.g2m_receivers], m2g_senders=inv[gr.m2g_senders],
            mesh_struct=gr.mesh_struct[perm], khop_rowptr=pm.indptr, khop_cols=pm.indices)
  kw = dict(num_layers=dims.num_layers, num_heads=dims.num_heads, attention="neighbour")
  np.testing.assert_allclose(O.denoiser_forward(params, gp, x, sigma, **kw),
                             O.denoiser_forward(params, gd, x, sigma, **kw), atol=1e-11)


def test_sampler_closed_form_linear_denoiser():
  """With F == 0 the preconditioned denoiser is D = c_skip*x, so every step is a
  scalar recursion that can be written down independently of the loop."""
  rng = np.random.default_rng(6)
  noise = rng.standard_normal((5, 1, 3))
  sig = O.noise_schedule(80.0, 0.03, 6, 7.0)
  cond = np.zeros((5, 1, 7))
  out, calls = O.dpm_solver_2s_sample(lambda f, s: np.zeros((5, 1, 3)), cond, np.arange(4, 7), noise, sig)
  assert calls == 12
  f = sig[0]
  for i in range(len(sig) - 1):
    s, sn = sig[i], sig[i + 1]
    sm = np.sqrt(s * sn)
    d = O.c_skip(max(s, 1e-6))
    mid = sm / s + (1 - sm / s) * d
    f = f * d if sn == 0 else f * (sn / s + (1 - sn / s) * O.c_skip(max(sm, 1e-6)) * mid)
  np.testing.assert_allclose(out, noise * f, rtol=1e-12)
  out2, calls2 = O.dpm_solver_2s_sample(lambda f, s: np.zeros((5, 1, 3)), cond, np.arange(4, 7), noise, sig, skip_dead_call=True)
  assert calls2 == 11
  np.testing.assert_array_equal(out, out2)


def test_sampler_matches_golden():
  gr, dims, params, x, _ = helpers.tiny_setup(batch=2)
  gd = helpers.graph_dict(gr)
  noise = np.random.default_rng(5).standard_normal((gr.num_grid_nodes, 2, dims.c_out))
  assert np.isclose(noise.sum(), GOLD["sampler_noise_sum"])
  net = lambda f, s: O.denoiser_forward(params, gd, f, s, num_layers=dims.num_layers,
                                        num_heads=dims.num_heads, attention="dense")
  out, calls = O.dpm_solver_2s_sample(net, x.astype(np.float64), GOLD["sampler_slots"], noise,
                                      GOLD["sampler_sigmas"], skip_dead_call=True)
  assert calls == int(GOLD["sampler_calls"]) - 1
  np.testing.assert_allclose(out, GOLD["sampler_out"], atol=1e-10)


def test_preconditioning_coefficients():
  s = np.array([0.03, 1.0, 80.0])
  np.testing.assert_allclose(O.c_in(s) ** 2 * (s ** 2 + 1), 1)
  np.testing.assert_allclose(O.c_out(s), s * O.c_in(s))
  np.testing.assert_allclose(O.c_skip(s) + O.c_out(s) ** 2, 1)
