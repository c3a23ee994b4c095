"""Per-tile chunk counts of the 1-degree attention launch and what the order of the tiles inside an XCD's range costs:
list-scheduling makespan (one workgroup per CU, 32 CUs per XCD) of the library's spatial order against longest-first.
usage: attention_tile_schedule.py [k_hop]      (needs a GPU: the mesh order is the library's)"""
import heapq, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from tests import helpers

k_hop = int(sys.argv[1]) if len(sys.argv) > 1 else 8
gr, dims, params, x, sigma = helpers.one_degree_setup(layers=1, k_hop=k_hop)
nd = helpers.make_native(gr, dims, params, 1)
perm = nd.debug_mesh_permutation()
nd.close()
M = gr.num_mesh_nodes
rowptr, cols = gr.khop_rowptr, gr.khop_cols
chunks = []
for t0 in range(0, M, 32):
  nodes = perm[t0:t0 + 32]
  u = np.unique(np.concatenate([cols[rowptr[n]:rowptr[n + 1]] for n in nodes]))
  chunks.append(-(-len(u) // 32))
chunks = np.array(chunks)
n = len(chunks)
print(f"k_hop {k_hop}: {n} tiles, chunks per tile min {chunks.min()} mean {chunks.mean():.2f} max {chunks.max()}; histogram {np.bincount(chunks)[chunks.min():].tolist()} from {chunks.min()}")
base, extra = n // 8, n % 8
def makespan(order_fn, fixed=1.3):
  worst, tot = 0.0, 0.0
  for xcd in range(8):
    lo = xcd * base + min(xcd, extra)
    cnt = base + (1 if xcd < extra else 0)
    jobs = order_fn(chunks[lo:lo + cnt]) + fixed          # + prologue in chunk units
    cus = [0.0] * 32
    heapq.heapify(cus)
    for j in jobs:
      heapq.heappush(cus, heapq.heappop(cus) + j)
    worst = max(worst, max(cus))
  return worst
print("makespan in chunk units: spatial order %.1f, longest first %.1f, shortest first %.1f; lower bound (total / 256) %.1f" % (
    makespan(lambda c: c.astype(float)), makespan(lambda c: np.sort(c)[::-1].astype(float)),
    makespan(lambda c: np.sort(c).astype(float)), (chunks.sum() + 1.3 * n) / 256))
