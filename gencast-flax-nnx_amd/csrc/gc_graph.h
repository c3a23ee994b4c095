// Host-side graph preparation for the HIP kernels (runs once in gc_set_graph).
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

namespace gc {

struct HostGraph {
  int G = 0, M = 0, E1 = 0, E2 = 0;
  // Internal mesh numbering: new id -> caller id, and its inverse.
  std::vector<int> perm, inv;
  // Edge lists with mesh ids renumbered.
  std::vector<int> g2m_snd, g2m_rcv, m2g_snd, m2g_rcv;
  // CSR by receiver (edge ids ascending inside a row).
  std::vector<int> g2m_ptr, g2m_eid, m2g_ptr, m2g_eid;
  // every grid node receives exactly 3 mesh2grid edges (the reference's construction): the mesh2grid edge arrays above
  // are then in the INTERNAL order "sorted by receiver" (row 3g + s), m2g_order[row] = the caller's edge id
  bool m2g_tri = false;
  std::vector<int> m2g_order;
  // Attention tiles: tile t = internal mesh nodes [32t, 32t+32).
  int n_tiles = 0;
  std::vector<int> tile_chunk_start;   // [n_tiles+1], in 32-key chunks
  std::vector<int> union_idx;          // [chunks*32] internal mesh ids (padding repeats a valid id)
  std::vector<unsigned> mask_bits;     // [chunks*32]: bit k of word (chunk, q) = key k attended by query q
  long long khop_nnz = 0;
};

// Returns "" on success, else an error message.
std::string build_host_graph(int G, int M, int E1, const int32_t* g2m_s, const int32_t* g2m_r, int E2,
                             const int32_t* m2g_s, const int32_t* m2g_r, const int32_t* khop_rowptr,
                             const int32_t* khop_cols, const float* mesh_xyz, HostGraph* out);

}  // namespace gc
