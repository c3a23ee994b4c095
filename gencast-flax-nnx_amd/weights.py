"""Weight naming, shapes, synthetic init and the flat weight file.

Names are the reference's Flax-NNX attribute paths:
`denoiser.noise_level_encoder.linear_{0,1}` (common/mlp.py:251),
`denoiser.predictor.{grid2mesh_gnn,mesh2grid_gnn}.{embedder_network,processor_networks.0.graph_network,decoder_network}...`
(common/deep_typed_graph_net.py:391-490; common/typed_graph_net.py:283-313) and
`denoiser.predictor.mesh_gnn.batch_first_transformer.blocks.<i>...`
(gencast/sparse_transformer.py:262-305,477-483,620-622).  Kernels are stored
`(in, out)` like flax (`y = x @ kernel + bias`).
"""
from __future__ import annotations

import dataclasses
from typing import Dict, Tuple

import numpy as np

P_NOISE = "denoiser.noise_level_encoder"
P_G2M = "denoiser.predictor.grid2mesh_gnn"
P_M2G = "denoiser.predictor.mesh2grid_gnn"
P_TR = "denoiser.predictor.mesh_gnn.batch_first_transformer"

COND_DIM = 16           # norm_conditioning_dim (deep_typed_graph_net.py:159,207)
STRUCT_NODE = 3         # (cos theta, cos phi, sin phi)
STRUCT_EDGE = 4         # (|d|, dx, dy, dz)


@dataclasses.dataclass(frozen=True)
class ModelDims:
  """Sizes that fix every weight shape."""
  c_in: int              # stacked inputs + forcings channels (262 for the nano task)
  c_out: int             # predicted channels (82)
  latent: int            # GNN latent = MLP hidden (latent_size)
  d_model: int
  num_heads: int
  ffw_hidden: int
  num_layers: int
  noise_num_frequencies: int = 32
  noise_hidden: int = 32
  hidden_layers: int = 1   # hidden layers of every GNN MLP (DenoiserArchitectureConfig.hidden_layers, denoiser.py:135)

  def __post_init__(self):
    if self.hidden_layers < 1:
      raise ValueError("hidden_layers must be >= 1")
    if self.latent != self.d_model:
      raise ValueError("mesh latent (latent_size) must equal transformer d_model "
                       "(the transformer consumes the grid2mesh mesh latents directly)")
    if self.d_model % self.num_heads:
      raise ValueError("num_heads has to divide d_model exactly")


def _mlp_specs(path, n_in, n_hidden, n_out, cond: bool, hidden_layers: int = 1):
  """common/mlp.py:152-203: `hidden_layers` x (Linear, activation) then the output Linear, as an nnx.Sequential:
  the i-th Linear is `layers.{2 i}`."""
  s = {}
  width = n_in
  for i in range(hidden_layers):
    s[f"{path}.network.network.layers.{2 * i}.kernel"] = (width, n_hidden)
    s[f"{path}.network.network.layers.{2 * i}.bias"] = (n_hidden,)
    width = n_hidden
  s[f"{path}.network.network.layers.{2 * hidden_layers}.kernel"] = (n_hidden, n_out)
  s[f"{path}.network.network.layers.{2 * hidden_layers}.bias"] = (n_out,)
  if cond:
    c = f"{path}.norm_conditioning_layer.conditional_linear_layer"
    s[f"{c}.kernel"] = (COND_DIM, 2 * n_out)
    s[f"{c}.bias"] = (2 * n_out,)
  return s


def param_specs(d: ModelDims) -> Dict[str, Tuple[int, ...]]:
  """name -> shape for every parameter on the sampling path."""
  L, D, F = d.latent, d.d_model, d.ffw_hidden
  s: Dict[str, Tuple[int, ...]] = {}
  s[f"{P_NOISE}.linear_0.kernel"] = (2 * d.noise_num_frequencies, d.noise_hidden)
  s[f"{P_NOISE}.linear_0.bias"] = (d.noise_hidden,)
  s[f"{P_NOISE}.linear_1.kernel"] = (d.noise_hidden, COND_DIM)
  s[f"{P_NOISE}.linear_1.bias"] = (COND_DIM,)
  node_in = STRUCT_NODE + d.c_in
  e = f"{P_G2M}.embedder_network"
  s.update(_mlp_specs(f"{e}.embed_edge_fns.grid2mesh", STRUCT_EDGE, L, L, True, d.hidden_layers))
  s.update(_mlp_specs(f"{e}.embed_node_fns.grid_nodes", node_in, L, L, True, d.hidden_layers))
  s.update(_mlp_specs(f"{e}.embed_node_fns.mesh_nodes", node_in, L, L, True, d.hidden_layers))
  g = f"{P_G2M}.processor_networks.0.graph_network"
  s.update(_mlp_specs(f"{g}.update_edge_fns.grid2mesh.edge_fn", 3 * L, L, L, True, d.hidden_layers))
  s.update(_mlp_specs(f"{g}.update_node_fns.grid_nodes.node_fn", L, L, L, True, d.hidden_layers))
  s.update(_mlp_specs(f"{g}.update_node_fns.mesh_nodes.node_fn", 2 * L, L, L, True, d.hidden_layers))
  s.update(_mlp_specs(f"{P_M2G}.embedder_network.embed_edge_fns.mesh2grid", STRUCT_EDGE, L, L, True, d.hidden_layers))
  g2 = f"{P_M2G}.processor_networks.0.graph_network"
  s.update(_mlp_specs(f"{g2}.update_edge_fns.mesh2grid.edge_fn", 3 * L, L, L, True, d.hidden_layers))
  s.update(_mlp_specs(f"{g2}.update_node_fns.grid_nodes.node_fn", 2 * L, L, L, True, d.hidden_layers))
  s.update(_mlp_specs(f"{P_M2G}.decoder_network.embed_node_fns.grid_nodes", L, L, d.c_out, False, d.hidden_layers))
  for i in range(d.num_layers):
    b = f"{P_TR}.blocks.{i}"
    for qkv in "qkv":
      s[f"{b}.attn_module.{qkv}_proj.linear.kernel"] = (D, D)
    s[f"{b}.attn_module.final_linear.kernel"] = (D, D)
    s[f"{b}.attn_module.final_linear.bias"] = (D,)
    s[f"{b}.ffw_module.mlp.layers.0.kernel"] = (D, F)
    s[f"{b}.ffw_module.mlp.layers.0.bias"] = (F,)
    s[f"{b}.ffw_module.mlp.layers.2.kernel"] = (F, D)
    s[f"{b}.ffw_module.mlp.layers.2.bias"] = (D,)
    for nc in ("norm_cond_attn", "norm_cond_ffw"):
      s[f"{b}.{nc}.conditional_linear_layer.kernel"] = (COND_DIM, 2 * D)
      s[f"{b}.{nc}.conditional_linear_layer.bias"] = (2 * D,)
  s[f"{P_TR}.final_norm_cond.conditional_linear_layer.kernel"] = (COND_DIM, 2 * D)
  s[f"{P_TR}.final_norm_cond.conditional_linear_layer.bias"] = (2 * D,)
  return s


# The reference also instantiates a mesh-node update in the mesh2grid GNN whose
# result is never read (denoiser.py:403-406,765-768); weight files may carry it.
IGNORED_PREFIXES = (
    f"{P_M2G}.processor_networks.0.graph_network.update_node_fns.mesh_nodes.",)


def random_params(d: ModelDims, seed: int = 3, dtype=np.float32) -> Dict[str, np.ndarray]:
  """Non-degenerate synthetic weights: kernels N(0, 1/fan_in)
  including conditioning and output projections, biases N(0, 0.1^2).

  The reference's own init zeroes the attention/FFW output projections and makes
  conditioning ~1e-8 (denoiser.py:93-95; mlp.py:44), which would hide most of the
  network from a parity test.
  """
  rng = np.random.default_rng(seed)
  out = {}
  for name, shape in param_specs(d).items():
    if name.endswith(".kernel"):
      w = rng.standard_normal(shape) / np.sqrt(shape[0])
    else:
      w = 0.1 * rng.standard_normal(shape)
    out[name] = w.astype(dtype)
  return out


def count_params(d: ModelDims) -> int:
  return int(sum(int(np.prod(s)) for s in param_specs(d).values()))


def save_params(path: str, params: Dict[str, np.ndarray]) -> None:
  """Flat weight file: an .npz keyed by the NNX paths above."""
  np.savez(path, **params)


def load_params(path: str) -> Dict[str, np.ndarray]:
  with np.load(path) as z:
    return {k: z[k] for k in z.files}


# ---- import of the reference's checkpoints ---------------------------------

def clean_state(obj):
  """The reference's checkpoint clean-up (training/evaluation.py:137-176) on plain containers: drop the
  normalisation-statistics datasets stored beside the model, hoist the contents of a 'graph_network'
  wrapper one level up, drop private buffers (keys starting with '_'), unwrap {0: leaf} singletons and
  one-element lists / tuples."""
  if isinstance(obj, dict):
    if "10m_u_component_of_wind" in obj or "2m_temperature" in obj:
      return None
    if "graph_network" in obj:
      return clean_state(obj["graph_network"])
    new = {}
    for k, v in obj.items():
      if isinstance(k, str) and k.startswith("_"):
        continue
      cv = clean_state(v)
      if cv is not None:
        new[k] = cv
    keys = list(new.keys())
    if len(keys) == 1 and keys[0] in (0, "0") and not isinstance(new[keys[0]], (dict, list, tuple)):
      return new[keys[0]]
    return new
  if isinstance(obj, (list, tuple)) and len(obj) == 1:
    return clean_state(obj[0])
  return obj


def flatten_state(state, prefix: str = "") -> Dict[str, np.ndarray]:
  """Nested dict / list of arrays -> {dotted NNX path: array}.  A trailing `.value` (how flax-nnx
  serialises a Variable) is dropped."""
  out: Dict[str, np.ndarray] = {}
  if isinstance(state, dict):
    items = state.items()
  elif isinstance(state, (list, tuple)):
    items = enumerate(state)
  else:
    out[prefix[:-1] if prefix.endswith(".") else prefix] = np.asarray(state)
    return out
  for k, v in items:
    if k == "value" and not isinstance(v, (dict, list, tuple)):
      out[prefix[:-1]] = np.asarray(v)
    else:
      out.update(flatten_state(v, f"{prefix}{k}."))
  return out


def import_reference_state(state, d: ModelDims, *, strict: bool = True) -> Dict[str, np.ndarray]:
  """A restored (orbax) state tree of the reference's GenCast module -> the flat weight dict
  `gc_load_weight` takes.  Applies `clean_state`, flattens, and reconciles the one naming ambiguity:
  `clean_state` removes the `graph_network` level that this build's names (= the live module's
  attribute paths, common/deep_typed_graph_net.py:414-490) keep, so a name is accepted with or
  without it.  Parameters the sampling path never reads (the dead mesh2grid mesh update, optimiser
  or loss state) are ignored; with `strict`, a missing or mis-shaped live parameter raises."""
  flat = flatten_state(clean_state(state) or {})
  specs = param_specs(d)
  strip = lambda n: n.replace(".graph_network.", ".")
  by_stripped = {strip(n): n for n in specs}
  out: Dict[str, np.ndarray] = {}
  for name, arr in flat.items():
    target = name if name in specs else by_stripped.get(strip(name))
    if target is None:
      continue
    a = np.asarray(arr, dtype=np.float32)
    if tuple(a.shape) != tuple(specs[target]):
      if strict:
        raise ValueError(f"{name}: shape {a.shape}, expected {specs[target]}")
      continue
    out[target] = a
  missing = sorted(set(specs) - set(out))
  if strict and missing:
    raise ValueError(f"{len(missing)} parameters missing from the checkpoint, e.g. {missing[:3]}")
  return out
