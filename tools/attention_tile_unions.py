"""Key-union sizes of the attention query tiles: for the library's internal mesh order (groups of 32 spatially compact
nodes), the mean number of distinct keys a 32-query tile attends to, and what 64-query tiles (two neighbouring
groups) would need -- the bytes the kernel gathers per query scale with union / tile size.
usage: attention_tile_unions.py [nano|one_degree]"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from tests import helpers

which = sys.argv[1] if len(sys.argv) > 1 else "one_degree"
gr, dims, params, x, sigma = helpers.nano_setup() if which == "nano" else helpers.one_degree_setup(layers=1)
nd = helpers.make_native(gr, dims, params, 1)
perm = nd.debug_mesh_permutation()            # perm[new] = caller id
stats = nd.debug_attention_stats()
nd.close()
M = gr.num_mesh_nodes
rowptr, cols = gr.khop_rowptr, gr.khop_cols
deg = np.diff(rowptr)
for T in (32, 64, 128):
  sizes = []
  for t0 in range(0, M, T):
    nodes = perm[t0:t0 + T]
    u = np.unique(np.concatenate([cols[rowptr[n]:rowptr[n + 1]] for n in nodes]))
    sizes.append(len(u))
  sizes = np.array(sizes)
  chunks = np.ceil(sizes / 32).sum()
  print(f"{which}: {T}-query tiles: {len(sizes)} tiles, mean key union {sizes.mean():.1f} (max {sizes.max()}), "
        f"keys gathered per query {sizes.sum() / M:.2f}, 32-key chunks {int(chunks)}; mean neighbourhood {deg.mean():.1f}")
print("library:", stats)
