// Host-side graph preparation: validation, locality renumbering of mesh nodes,
// CSR-by-receiver for the deterministic segment sums, and the attention tiles.
//
// The reference makes its attention tractable by RCM-banding the mesh and
// computing dense 3-block-wide logits (gencast/denoiser.py:849-867;
// gencast/sparse_transformer.py:555-567).  Mesh numbering is internal to the
// model (weights are shared by all nodes), so this build instead renumbers the
// mesh by recursive coordinate bisection into spatially compact groups of 32
// nodes.  A compact 32-query tile's neighbourhoods overlap heavily: the union of
// its k-hop sets is ~2x one set, which is what lets the attention kernel reuse
// every gathered K/V row across 32 queries at ~50 % mask density.
#include "gc_graph.h"

#include <algorithm>
#include <numeric>

namespace gc {

namespace {

void rcb_split(const float* xyz, std::vector<int>& ids, int begin, int end, int tile) {
  const int n = end - begin;
  if (n <= tile) return;
  const int n_tiles = (n + tile - 1) / tile;
  const int left_tiles = n_tiles / 2;
  const int left = left_tiles * tile;  // every tile but the very last stays full
  float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
  for (int i = begin; i < end; ++i)
    for (int a = 0; a < 3; ++a) {
      const float v = xyz[3 * (size_t)ids[i] + a];
      lo[a] = std::min(lo[a], v);
      hi[a] = std::max(hi[a], v);
    }
  int axis = 0;
  for (int a = 1; a < 3; ++a)
    if (hi[a] - lo[a] > hi[axis] - lo[axis]) axis = a;
  std::stable_sort(ids.begin() + begin, ids.begin() + end, [&](int p, int q) {
    return xyz[3 * (size_t)p + axis] < xyz[3 * (size_t)q + axis];
  });
  rcb_split(xyz, ids, begin, begin + left, tile);
  rcb_split(xyz, ids, begin + left, end, tile);
}

void csr_by_receiver(const std::vector<int>& rcv, int n_nodes, std::vector<int>* ptr,
                     std::vector<int>* eid) {
  ptr->assign(n_nodes + 1, 0);
  for (int r : rcv) (*ptr)[r + 1]++;
  for (int i = 0; i < n_nodes; ++i) (*ptr)[i + 1] += (*ptr)[i];
  eid->resize(rcv.size());
  std::vector<int> cur(ptr->begin(), ptr->end() - 1);
  for (int e = 0; e < (int)rcv.size(); ++e) (*eid)[cur[rcv[e]]++] = e;  // ascending edge id per row
}

}  // namespace

std::string build_host_graph(int G, int M, int E1, const int32_t* g2m_s, const int32_t* g2m_r, int E2,
                             const int32_t* m2g_s, const int32_t* m2g_r, const int32_t* khop_rowptr,
                             const int32_t* khop_cols, const float* mesh_xyz, HostGraph* out) {
  if (G <= 0 || M <= 0 || E1 <= 0 || E2 <= 0) return "graph sizes must be positive";
  for (int e = 0; e < E1; ++e)
    if (g2m_s[e] < 0 || g2m_s[e] >= G || g2m_r[e] < 0 || g2m_r[e] >= M)
      return "grid2mesh edge index out of range";
  for (int e = 0; e < E2; ++e)
    if (m2g_s[e] < 0 || m2g_s[e] >= M || m2g_r[e] < 0 || m2g_r[e] >= G)
      return "mesh2grid edge index out of range";
  if (khop_rowptr[0] != 0) return "khop_rowptr[0] must be 0";
  for (int i = 0; i < M; ++i) {
    if (khop_rowptr[i + 1] < khop_rowptr[i]) return "khop_rowptr must be non-decreasing";
    bool self = false;
    for (int p = khop_rowptr[i]; p < khop_rowptr[i + 1]; ++p) {
      if (khop_cols[p] < 0 || khop_cols[p] >= M) return "khop column out of range";
      self |= (khop_cols[p] == i);
    }
    if (!self) return "every khop row must contain its own node (self edge)";
  }

  HostGraph& g = *out;
  g = HostGraph();
  g.G = G; g.M = M; g.E1 = E1; g.E2 = E2;
  g.khop_nnz = khop_rowptr[M];

  g.perm.resize(M);
  std::iota(g.perm.begin(), g.perm.end(), 0);
  if (mesh_xyz) rcb_split(mesh_xyz, g.perm, 0, M, 32);
  g.inv.resize(M);
  for (int i = 0; i < M; ++i) g.inv[g.perm[i]] = i;

  g.g2m_snd.assign(g2m_s, g2m_s + E1);
  g.g2m_rcv.resize(E1);
  for (int e = 0; e < E1; ++e) g.g2m_rcv[e] = g.inv[g2m_r[e]];
  g.m2g_snd.resize(E2);
  for (int e = 0; e < E2; ++e) g.m2g_snd[e] = g.inv[m2g_s[e]];
  g.m2g_rcv.assign(m2g_r, m2g_r + E2);
  csr_by_receiver(g.g2m_rcv, M, &g.g2m_ptr, &g.g2m_eid);
  csr_by_receiver(g.m2g_rcv, G, &g.m2g_ptr, &g.m2g_eid);
  // mesh2grid: the reference gives every grid node exactly the 3 vertices of its containing triangle, as edges
  // 3g, 3g+1, 3g+2 (common/grid_mesh_connectivity.py:118-131).  Edge numbering is as internal as mesh numbering, so when
  // every grid node has in-degree 3 the edge set is kept SORTED BY RECEIVER (ascending caller edge id inside a triple --
  // the order jraph.segment_sum adds them in): internal row 3g + s.  The fused edge update then sums a triple in its
  // epilogue and never stores the updated edges (gc_mlp_ws_kernel, MlpArgs::tri).
  g.m2g_tri = (E2 == 3 * G);
  for (int i = 0; g.m2g_tri && i < G; ++i) g.m2g_tri = (g.m2g_ptr[i + 1] - g.m2g_ptr[i] == 3);
  g.m2g_order.resize(E2);
  if (g.m2g_tri) {
    g.m2g_order = g.m2g_eid;                                  // internal row -> caller edge id
    std::vector<int> snd(E2), rcv(E2);
    for (int i = 0; i < E2; ++i) { snd[i] = g.m2g_snd[g.m2g_order[i]]; rcv[i] = g.m2g_rcv[g.m2g_order[i]]; }
    g.m2g_snd.swap(snd);
    g.m2g_rcv.swap(rcv);
    std::iota(g.m2g_eid.begin(), g.m2g_eid.end(), 0);         // the CSR of the internal order is the identity
  } else {
    std::iota(g.m2g_order.begin(), g.m2g_order.end(), 0);
  }

  // Attention tiles.
  g.n_tiles = (M + 31) / 32;
  g.tile_chunk_start.assign(g.n_tiles + 1, 0);
  std::vector<int> uni;
  for (int t = 0; t < g.n_tiles; ++t) {
    const int q0 = t * 32, q1 = std::min(M, q0 + 32);
    uni.clear();
    for (int q = q0; q < q1; ++q) {
      const int old = g.perm[q];
      for (int p = khop_rowptr[old]; p < khop_rowptr[old + 1]; ++p) uni.push_back(g.inv[khop_cols[p]]);
    }
    std::sort(uni.begin(), uni.end());
    uni.erase(std::unique(uni.begin(), uni.end()), uni.end());
    const int n_chunks = ((int)uni.size() + 31) / 32;
    const int base = g.tile_chunk_start[t];
    g.tile_chunk_start[t + 1] = base + n_chunks;
    g.union_idx.resize((size_t)(base + n_chunks) * 32, uni.back());
    g.mask_bits.resize((size_t)(base + n_chunks) * 32, 0u);
    for (size_t i = 0; i < uni.size(); ++i) g.union_idx[(size_t)base * 32 + i] = uni[i];
    for (size_t i = uni.size(); i < (size_t)n_chunks * 32; ++i)
      g.union_idx[(size_t)base * 32 + i] = uni.back();
    for (int q = q0; q < q1; ++q) {
      const int old = g.perm[q];
      for (int p = khop_rowptr[old]; p < khop_rowptr[old + 1]; ++p) {
        const int key = g.inv[khop_cols[p]];
        const int pos = (int)(std::lower_bound(uni.begin(), uni.end(), key) - uni.begin());
        g.mask_bits[(size_t)(base + pos / 32) * 32 + (q - q0)] |= (1u << (pos % 32));
      }
    }
  }
  return "";
}

}  // namespace gc
