import sys, numpy as np
sys.path.insert(0, '.')
from oracle import gencast_oracle as O
from tests import helpers
F16=np.float16
for batch in (2,1):
  gr, dims, params, x, sigma = helpers.tiny_setup(batch=batch, seed=31)
  nd = helpers.make_native(gr, dims, params, batch)
  nd.set_option("features","f16")
  nd.denoise(x, sigma)
  gd = helpers.graph_dict(gr)
  p64 = {k: np.asarray(v, np.float64) for k, v in params.items()}
  cond = O.noise_level_encoding(p64, sigma.astype(np.float64))
  B, M, D = batch, gr.num_mesh_nodes, dims.latent
  G = gr.num_grid_nodes
  f = {k: nd.debug_fetch(k).astype(np.float64) for k in ("g0","m0","e1","g1","m2","f1","e0_hat","f0_hat","agg2","g2")}
  bc = lambda arr: np.broadcast_to(np.asarray(arr, np.float64)[:, None, :], (arr.shape[0], B, arr.shape[1]))
  def hat_to(name, hat):
    emb = f"{name}.norm_conditioning_layer.conditional_linear_layer"
    so = O._cond_linear(cond, p64[f"{emb}.kernel"], p64[f"{emb}.bias"])
    return O._R(bc(hat) * (so[None,:,:D]+1.0) + so[None,:,D:])
  for tag, hatn, embn, fn, snd, rcv, sn, rn, outn in (
      ("g2m edge", "e0_hat", f"{O.P_G2M}.embedder_network.embed_edge_fns.grid2mesh", f"{O.P_G2M}.processor_networks.0.graph_network.update_edge_fns.grid2mesh.edge_fn", gd["g2m_senders"], gd["g2m_receivers"], "g0", "m0", "e1"),
      ("m2g edge", "f0_hat", f"{O.P_M2G}.embedder_network.embed_edge_fns.mesh2grid", f"{O.P_M2G}.processor_networks.0.graph_network.update_edge_fns.mesh2grid.edge_fn", gd["m2g_senders"], gd["m2g_receivers"], "m2", "g1", "f1")):
    S = f[sn].reshape(-1, B, D); R = f[rn].reshape(-1, B, D)
    with O.feature_rounding(F16):
      e0 = hat_to(embn, f[hatn])
      inp = np.concatenate([e0, S[snd], R[rcv]], -1)
      ref = O.mlp_norm_cond(p64, fn, inp, cond)
      # stage internals
      h = O._R(O.swish(O._linear_f32(inp, p64[f"{fn}.network.network.layers.0.kernel"], p64[f"{fn}.network.network.layers.0.bias"])))
      y2 = O.linear(h, p64[f"{fn}.network.network.layers.2.kernel"], p64[f"{fn}.network.network.layers.2.bias"])
    got = f[outn].reshape(ref.shape)
    eq = (got == ref)
    print(batch, tag, "match", eq.mean(), "per batch", [eq[:, b].mean() for b in range(B)], "rows all-equal frac", eq.all(-1).mean())
    rowm = eq.mean(-1)
    print("   worst rows", np.sort(rowm.ravel())[:5], "row match quantiles", np.quantile(rowm, [0.01, 0.1, 0.5]))
    print("   |y2| stats: max", np.abs(y2).max(), "row std min/median", np.quantile(y2.std(-1), [0, 0.5]), " |h| max", np.abs(h).max())
    # column structure
    print("   col-block match", [round(float(eq[..., c:c+32].mean()),4) for c in range(0, D, 32)])
  # node update m2g grid
  gn2 = f"{O.P_M2G}.processor_networks.0.graph_network"
  g1 = f["g1"].reshape(G,B,D); agg2=f["agg2"].reshape(G,B,D)
  with O.feature_rounding(F16):
    g2 = O._R(g1 + O.mlp_norm_cond(p64, f"{gn2}.update_node_fns.grid_nodes.node_fn", np.concatenate([g1, agg2], -1), cond))
  print(batch, "m2g grid update match", (f["g2"].reshape(g2.shape)==g2).mean())
  nd.close()
