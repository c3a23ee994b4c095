// C ABI of libgencast_hip.so (include/gencast_hip.h): handle, weight registry,
// graph upload, the denoiser forward and the DPM-Solver++2S loop.
#include <hip/hip_runtime.h>

#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>   // types and enums only: librccl.so.1 is dlopen'ed on first use (gc_comm_*)

#include "../../include/gencast_hip.h"
#include "../../include/gencast_hip_debug.h"
#include "gc_graph.h"
#include "gc_kernels.h"

// gc_a16 = the same kernels compiled a second time (gc_kernels.hip with -DGC_TU_A16): 2 MFMAs per product for
// exact-fp16 activation operands.  Its argument structs are the same declarations in another namespace.
template <class To, class From>
static const To& a16_view(const From& v) {
  static_assert(sizeof(To) == sizeof(From), "argument struct mismatch between the two kernel builds");
  return reinterpret_cast<const To&>(v);
}

namespace {

thread_local std::string g_create_error;

const char* P_NOISE = "denoiser.noise_level_encoder";
const char* P_G2M = "denoiser.predictor.grid2mesh_gnn";
const char* P_M2G = "denoiser.predictor.mesh2grid_gnn";
const char* P_TR = "denoiser.predictor.mesh_gnn.batch_first_transformer";

int round_up(int v, int m) { return (v + m - 1) / m * m; }

struct DevMlp {        // device-side layout of one MLPWithNormConditioning
  float* w1t = nullptr; int ldw1 = 0;
  float* b1 = nullptr;
  float* w2t = nullptr;
  float* b2 = nullptr;
  float *w1s = nullptr, *w2s = nullptr;   // S16 (split-fp16) encodings of w1t / w2t
  float *w1f = nullptr, *w2f = nullptr;   // WF16 (MFMA fragment order) images; w1f's K is padded to k1f
  float *w1x = nullptr, *w2x = nullptr, *w1e_x = nullptr;   // WF32 images (exact-f32 family on the weight-streaming form)
  int k1f = 0;
  // edge MLPs only: first layer split by input block [e | sender | receiver] (each L rows of W1)
  float *w1e_t = nullptr, *w1e_s = nullptr;   // [hidden][L]  edge block
  float *w1snd_t = nullptr, *w1snd_s = nullptr, *w1rcv_t = nullptr, *w1rcv_s = nullptr;
  float *w1e_f = nullptr, *w1snd_f = nullptr, *w1rcv_f = nullptr;   // WF16 images of the three blocks (K = L)
  int k1e = 0, ldw1e = 0;   // K (padded to 64) and row stride of the w1e_* images when they are NOT an edge block of width L
                            // (the noisy-columns block of the grid embedding: build_embed_cache)
  int n_out = 0, n_out_pad = 0;
  int cond_off = -1;   // offset of [scale | offset] in the conditioning buffer
  // hidden_layers >= 2 (common/mlp.py:166-183): the leading (Linear -> activation) layers, each run as a launch of
  // the same fused kernel with an identity second layer and no LayerNorm; this struct then holds the LAST hidden
  // Linear as its first layer and the output Linear as its second
  std::vector<DevMlp> pre;
};

struct DevLayer {      // one transformer block
  float* wqkv_t = nullptr;  // [3D][D]
  float* wo_t = nullptr;    // [D][D]
  float* bo = nullptr;
  float* w1_t = nullptr;    // [F][D]
  float* b1 = nullptr;
  float* w2_t = nullptr;    // [D][F]
  float* b2 = nullptr;
  // S16 (split-fp16) encodings of the same four matrices (f16x3 precision mode)
  float *wqkv_s = nullptr, *wo_s = nullptr, *w1_s = nullptr, *w2_s = nullptr;
  float *wqkv_f = nullptr, *wo_f = nullptr, *w1_f = nullptr, *w2_f = nullptr;   // WF16 fragment order
  float *wqkv_x = nullptr, *wo_x = nullptr, *w1_x = nullptr, *w2_x = nullptr;   // WF32: the exact-f32 family's fragment order
  int cond_attn = -1, cond_ffw = -1;
};

}  // namespace

struct gc_handle {
  gc_config cfg{};
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;             // side stream of forward() (grid-node update beside the mesh path)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool side_stream = false;                  // GC_TUNE_SIDE_STREAM=1 (measured slower: see forward())
  bool fuse_noisy = true;                    // GC_TUNE_FUSE_NOISY=0: the sampler writes the noisy slots with a launch of its own
  std::string err;
  bool has_graph = false, finalized = false, has_slots = false, has_cond = false, has_noise = false;
  bool finalized_weights = false;            // gc_finalize ran on the weights now loaded (gc_load_weight clears it)
  gc::HostGraph hg;

  std::map<std::string, std::vector<int64_t>> specs;  // expected shapes
  std::map<std::string, std::vector<float>> weights;  // host copies as loaded
  std::vector<void*> allocs;                          // everything hipMalloc'ed (freed in destroy)

  // graph (device)
  int *d_g2m_snd = nullptr, *d_g2m_rcv = nullptr, *d_m2g_snd = nullptr, *d_m2g_rcv = nullptr;
  int *d_g2m_ptr = nullptr, *d_g2m_eid = nullptr, *d_m2g_ptr = nullptr, *d_m2g_eid = nullptr;
  int *d_tile_start = nullptr, *d_union = nullptr;
  // work-item list of the attention launch (1.25 rounds of tiles -> one round of whole tiles + one round of pieces):
  // build_attention_items; GC_TUNE_ATTN_ITEMS=0 switches it off
  int *d_att_items = nullptr, *d_att_tiles = nullptr;
  int att_n_items = 0;
  int last_att_items = 0;                    // work items of the last forward's attention launches (0: plain (tile, split) launch)
  unsigned* d_mask = nullptr;
  float *d_grid_struct = nullptr, *d_mesh_struct16 = nullptr, *d_e1_struct16 = nullptr,
        *d_e2_struct16 = nullptr;

  // weights (device)
  DevMlp g2m_embed_grid, g2m_embed_mesh, g2m_embed_edge, g2m_edge, g2m_mesh, g2m_grid;
  DevMlp m2g_embed_edge, m2g_edge, m2g_grid, m2g_dec;
  std::vector<DevLayer> layers;
  int cond_final = -1;
  int cond_total = 0;
  std::map<std::string, std::pair<int, int>> cond_sites;   // conditioning linear (parameter path) -> (offset, width) in d_cond
  float *d_nw0t = nullptr, *d_nb0 = nullptr, *d_nw1t = nullptr, *d_nb1 = nullptr;
  float *d_wc_all = nullptr, *d_bc_all = nullptr;

  // static embeddings (LayerNorm output, before conditioning)
  float *d_m0_hat = nullptr, *d_e0_hat = nullptr, *d_f0_hat = nullptr;

  // activations
  int kp = 0;
  float *d_sigma = nullptr, *d_condvec = nullptr, *d_cond = nullptr;
  float* d_cond_all = nullptr;               // sampler: conditioning of every call of the sample [calls][B][total]
  size_t cond_all_cap = 0;
  const float* cond_cur = nullptr;           // conditioning vectors the current forward() reads
  float *d_feats = nullptr, *d_xp = nullptr, *d_g0 = nullptr, *d_g1 = nullptr, *d_m0 = nullptr,
        *d_x = nullptr, *d_e1 = nullptr, *d_agg1 = nullptr, *d_qkv = nullptr, *d_att = nullptr,
        *d_u = nullptr, *d_m2 = nullptr, *d_f1 = nullptr, *d_agg2 = nullptr, *d_g2 = nullptr,
        *d_y = nullptr, *d_h = nullptr, *d_part = nullptr, *d_apart_o = nullptr, *d_apart_ml = nullptr,
        *d_pg = nullptr, *d_pm = nullptr;     // per-node first-layer products of the edge MLPs
  bool mlp_ws = true;                        // GC_TUNE_MLP_WS=0: LDS-staged MLP kernel
  // Grid embedding with its per-sample-constant part cached (dpm_solver_plus_plus_2s.py:107-112,
  // denoiser.py:654-659): inside one sample only the c_out noisy-target channels of the packed grid input change from
  // call to call.  At the start of a sample P = W1[static rows]^T [struct | inputs | forcings] is computed once
  // ([G B, L] float32, the noisy columns' weights zeroed); each call's embedding MLP then multiplies only the compact
  // noisy array xn [G B, c_out padded to 32] and adds P next to the bias (the add-term path of the split edge MLPs).
  bool mlp_pair = true;                      // GC_TUNE_MLP_PAIR=0: the grid2mesh edge update and the grid-node update as two launches
  bool embed_cache = true;                   // GC_TUNE_EMBED_CACHE=0: every call multiplies all 3 + c_in columns
  bool embed_cache_ready = false;            // the split weight images below match the current weights and slots
  bool embed_cache_live = false;             // inside a sample that runs on the cache: forward() reads d_xn / d_pstat
  int nwp = 0;                               // noisy columns padded to a multiple of 32
  float *d_xn = nullptr, *d_pstat = nullptr;
  float *w1st_t = nullptr, *w1st_s = nullptr;   // static first layer [L][kp]: float32 and S16, noisy columns zero
  DevMlp g2m_embed_grid_n;                   // g2m_embed_grid with the noisy-columns block as its first layer (+ P as add term)
  std::vector<int> h_slots;
  int64_t embed_cache_samples = 0;           // samples that ran on the cache (gc_get_counter "embed_cache")
  bool mlp_ws512 = true;                     // GC_TUNE_MLP_WS512=0: latent 512 on the LDS-staged MLP kernel
  bool m2g_fuse_sum = true;                  // GC_TUNE_M2G_FUSE_SUM=0: mesh2grid edge update + a segment-sum launch (the form every
                                             // graph with other in-degrees than 3 takes anyway)
  bool last_m2g_fused = false;               // the last forward summed the mesh2grid triples inside the edge MLP: f1 was not stored
  float *d_ones = nullptr, *d_zeros = nullptr;   // identity affine for gc_mlp_ws
  int hidden_layers = 1;                         // gc_set_option("hidden_layers"): hidden layers of every GNN MLP (denoiser.py:135)
  float* d_mlp_tmp[2] = {nullptr, nullptr};      // hidden_layers >= 2: [max rows][latent] hand-over between the launches of one MLP
  // autoregressive context update (gc_rollout_plan / gc_rollout_advance)
  float *d_feats2 = nullptr, *d_ro_a = nullptr, *d_ro_b = nullptr, *d_ro_forc = nullptr;
  int *d_ro_kind = nullptr, *d_ro_src = nullptr, *d_ro_sidx = nullptr;
  int ro_nforc = -1, ro_forc_cap = 0;
  bool has_sample = false;
  int max_tile_chunks = 0;                   // largest number of 32-key chunks of any attention tile
  int ffw_fused_slabs = 0;                   // > 0: gc_ffw_fused with this many hidden slices (= slabs)
  int ws_mt = 0;                             // GC_TUNE_WS_MT: force 32- (1) or 64-row (2) tiles
  bool attn_f16 = true;                      // GC_TUNE_ATTN_F16=0: f32-MFMA attention also in f16x3 mode
  bool attn_v2 = true;                       // GC_TUNE_ATTN_V2=0 (A/B): plain QKV epilogue + the exact-f32 attention kernel
  void* d_kv16 = nullptr;                    // K / V as fp16 hi / lo planes, written by the QKV projection
  bool attn_v2_force = false;
  bool kv16_live = false;                    // the last forward's K / V live in d_kv16 only (not in d_qkv)
  bool fuse_outrow = true;                   // GC_TUNE_FUSE_OUTROW=0: split-K out-projection + separate row pass
  bool gemm_ws = true;                       // GC_TUNE_GEMM_WS=0: LDS-staged f16x3 GEMM
  bool f32_ws = true;                        // exact-f32 family on the weight-streaming / fused kernels (WF32 images); GC_TUNE_F32_WS=0: LDS-staged GEMMs
  int ffw_xcd = 0;                           // GC_TUNE_FFW_XCD=1 (experiment): fused-FFW slices of a row tile + its row pass on one XCD
  bool fuse_combine = true;                  // GC_TUNE_FUSE_COMBINE=0: separate gc_attn_combine launch
  bool split_edge = false;                   // GC_TUNE_SPLIT_EDGE=1 enables the split edge MLPs
  // launch geometry (defaults chosen in gc_set_graph; GC_TUNE_* env vars override for experiments)
  int attn_splits = 1, out_splits = 1, ffw2_splits = 1;
  bool a16 = true;                           // GC_TUNE_A16=0: fp16-feature mode on float32 containers with 3 MFMAs per product (A/B, bit-identical)
  bool st16 = false;                         // the launches being enqueued use PHYSICAL fp16 activation storage (store16_ok)
  bool last_st16 = false;                    // ... and so did the last forward (gc_debug_fetch converts)
  int wt_stores = 0;                         // GC_TUNE_WT_STORES bit mask: 1 FFW slabs, 2 fused-MLP outputs (write-through stores)
  int mt_qkv = 1, mt_out = 1, mt_ffw1 = 1, mt_ffw2 = 1;
  // sampler state
  int* d_slots = nullptr;
  float *d_sx = nullptr, *d_sden = nullptr, *d_smid = nullptr, *d_noise = nullptr;
  // The initial noise is double-buffered: a resident sample whose f16x3 domain check is still pending may have to be
  // re-run from ITS noise, so an upload / draw for the next member that arrives before the check is resolved goes to
  // the other buffer (d_noise always = the buffer the next sample will read; last_noise = the pending sample's).
  float *d_noise_alt = nullptr, *last_noise = nullptr;
  float* d_stash = nullptr;            // gc_stash_sample: snapshot of a sample, downloaded on the side stream
  hipEvent_t ev_stash = nullptr;
  bool has_stash = false;

  // spherical white noise on the device + stochastic churn (gc_noise_*, gc_set_churn)
  int nz_L = 0, nz_lat = 0, nz_lon = 0;
  float *d_nz_leg = nullptr, *d_nz_cos = nullptr, *d_nz_sin = nullptr, *d_nz_coef = nullptr, *d_nz_f = nullptr;
  unsigned long long nz_key = 0, nz_stream = 0;
  std::vector<float> churn_rates;      // per solver step; empty = no churn
  float churn_inflation = 1.0f;

  // HIP-graph replay of the sampler (gc_set_option "graphs"): one captured graph per sample signature
  struct SampleGraph {
    std::vector<float> sigmas;
    int skip_dead = 1;
    const float* noise = nullptr;      // the initial-noise buffer baked into the graph (it is double-buffered)
    bool f16 = false, feat16 = false, st16 = false;
    hipGraph_t graph = nullptr;
    // null until the signature has been seen twice; TWO executables of the one captured graph, launched alternately,
    // each with an event recorded behind its last launch: an executable is never launched while its previous
    // instance may still be running (the host waits for that event first, outside any lock)
    hipGraphExec_t exec = nullptr, exec2 = nullptr;
    hipEvent_t done[2] = {nullptr, nullptr};
    int next = 0;
    int calls = 0;
    int64_t launches_per_call = 0, launches = 0;
    uint64_t last_use = 0;
  };
  bool use_graphs = true;              // GC_TUNE_GRAPH=0 / gc_set_option(h, "graphs", "off"): always enqueue eagerly
  std::vector<SampleGraph> sample_graphs;
  uint64_t graph_clock = 0;
  int64_t graph_replays = 0, graph_captures = 0;
  int debug_layer_limit = -1;  // gc_debug_set_layer_limit
  int debug_stop_layer = -1, debug_stop_phase = -1;   // gc_debug_set_stop: forward() returns inside this block
  bool f16x3 = true;           // GEMM-shaped kernels run as 3 fp16 MFMAs per product (gc_set_option)
  float g2m_agg_norm = 0.f;    // "grid2mesh_aggregate_normalization": the grid2mesh edge sums are divided by it (0: not)
  bool feat16 = false;         // "features" = "f16": activations rounded to fp16 where stored (BASELINE configs[4])
  // f16x3 domain guard: operands outside fp16 range poison the output with
  // NaN / Inf (no clamp anywhere); the output is checked on the device once per call and a poisoned
  // call is re-run on the exact-f32 kernels, which treat NaN / Inf / huge inputs like the reference.
  bool weights_f16_unsafe = false;   // a weight is non-finite or beyond fp16 range: f32 kernels only
  bool in_fallback = false;          // forward() is running the f32 re-run of a poisoned call
  unsigned* d_nonfinite = nullptr;   // device counter bumped by gc_finite_check
  unsigned* h_nonfinite = nullptr;   // pinned host copy
  unsigned nonfinite_seen = 0;
  int64_t range_fallbacks = 0;       // calls re-run in f32 (gc_get_counter "range_fallbacks")
  bool guard_pending = false;        // a resident sample has not been checked yet
  std::vector<float> last_sigmas;    // arguments of that sample, for the re-run
  int last_skip_dead = 1;
  unsigned long long last_stream0 = 0;
  int64_t launches_last_call = 0, launch_count = 0;   // kernel launches of the last denoiser forward
  // pinned staging buffers of the asynchronous uploads (caller buffers are free on return)
  float *pin_cond = nullptr, *pin_noise = nullptr, *pin_forc = nullptr;
  size_t pin_forc_cap = 0;
  hipEvent_t ev_pin = nullptr;       // last H2D copy out of a staging buffer
  // ensemble exchange (gc_comm_*): one RCCL communicator per handle, collectives on h->stream
  ncclComm_t comm = nullptr;
  int comm_rank = 0, comm_world = 1;
  double* d_comm_scalar = nullptr;

  // profiling
  int prof_cls = -1;
  int prof_stride = 1;
  unsigned prof_seen = 0;
  std::vector<hipEvent_t> prof_events;
  size_t prof_used = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

namespace {

#define GC_HIP(h, call)                                                                     \
  do {                                                                                      \
    hipError_t e__ = (call);                                                                \
    if (e__ != hipSuccess) {                                                                \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(e__);                        \
      return GC_ERR_HIP;                                                                    \
    }                                                                                       \
  } while (0)

int fail(gc_handle* h, int code, const std::string& msg) {
  h->err = msg;
  return code;
}

template <typename T>
int dev_alloc(gc_handle* h, T** p, size_t count) {
  void* q = nullptr;
  GC_HIP(h, hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)));
  h->allocs.push_back(q);
  *p = reinterpret_cast<T*>(q);
  return GC_OK;
}

template <typename T>
int dev_upload(gc_handle* h, T** p, const std::vector<T>& v) {
  int rc = dev_alloc(h, p, v.size());
  if (rc) return rc;
  if (!v.empty()) GC_HIP(h, hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return GC_OK;
}

void add_mlp_specs(gc_handle* h, const std::string& p, int n_in, int n_hid, int n_out, bool cond) {
  // common/mlp.py:166-199: hidden_layers x (Linear, activation), then the output Linear; nnx.Sequential index 2 i = i-th Linear
  const int nh = h->hidden_layers;
  for (int i = 0; i < nh; ++i) {
    const std::string l = p + ".network.network.layers." + std::to_string(2 * i);
    h->specs[l + ".kernel"] = {i == 0 ? n_in : n_hid, n_hid};
    h->specs[l + ".bias"] = {n_hid};
  }
  const std::string l = p + ".network.network.layers." + std::to_string(2 * nh);
  h->specs[l + ".kernel"] = {n_hid, n_out};
  h->specs[l + ".bias"] = {n_out};
  if (cond) {
    h->specs[p + ".norm_conditioning_layer.conditional_linear_layer.kernel"] = {gc::kCondDim, 2 * n_out};
    h->specs[p + ".norm_conditioning_layer.conditional_linear_layer.bias"] = {2 * n_out};
  }
}

void build_specs(gc_handle* h) {
  const gc_config& c = h->cfg;
  const int L = c.latent_size, D = c.d_model, F = c.ffw_hidden;
  const std::string n = P_NOISE, g = P_G2M, m = P_M2G, t = P_TR;
  h->specs[n + ".linear_0.kernel"] = {2 * c.noise_num_frequencies, c.noise_hidden};
  h->specs[n + ".linear_0.bias"] = {c.noise_hidden};
  h->specs[n + ".linear_1.kernel"] = {c.noise_hidden, gc::kCondDim};
  h->specs[n + ".linear_1.bias"] = {gc::kCondDim};
  const int node_in = 3 + c.c_in;
  add_mlp_specs(h, g + ".embedder_network.embed_edge_fns.grid2mesh", 4, L, L, true);
  add_mlp_specs(h, g + ".embedder_network.embed_node_fns.grid_nodes", node_in, L, L, true);
  add_mlp_specs(h, g + ".embedder_network.embed_node_fns.mesh_nodes", node_in, L, L, true);
  const std::string gn = g + ".processor_networks.0.graph_network";
  add_mlp_specs(h, gn + ".update_edge_fns.grid2mesh.edge_fn", 3 * L, L, L, true);
  add_mlp_specs(h, gn + ".update_node_fns.grid_nodes.node_fn", L, L, L, true);
  add_mlp_specs(h, gn + ".update_node_fns.mesh_nodes.node_fn", 2 * L, L, L, true);
  add_mlp_specs(h, m + ".embedder_network.embed_edge_fns.mesh2grid", 4, L, L, true);
  const std::string gn2 = m + ".processor_networks.0.graph_network";
  add_mlp_specs(h, gn2 + ".update_edge_fns.mesh2grid.edge_fn", 3 * L, L, L, true);
  add_mlp_specs(h, gn2 + ".update_node_fns.grid_nodes.node_fn", 2 * L, L, L, true);
  add_mlp_specs(h, m + ".decoder_network.embed_node_fns.grid_nodes", L, L, c.c_out, false);
  for (int i = 0; i < c.num_layers; ++i) {
    const std::string b = t + ".blocks." + std::to_string(i);
    for (const char* q : {"q", "k", "v"})
      h->specs[b + ".attn_module." + q + "_proj.linear.kernel"] = {D, D};
    h->specs[b + ".attn_module.final_linear.kernel"] = {D, D};
    h->specs[b + ".attn_module.final_linear.bias"] = {D};
    h->specs[b + ".ffw_module.mlp.layers.0.kernel"] = {D, F};
    h->specs[b + ".ffw_module.mlp.layers.0.bias"] = {F};
    h->specs[b + ".ffw_module.mlp.layers.2.kernel"] = {F, D};
    h->specs[b + ".ffw_module.mlp.layers.2.bias"] = {D};
    for (const char* nc : {"norm_cond_attn", "norm_cond_ffw"}) {
      h->specs[b + "." + nc + ".conditional_linear_layer.kernel"] = {gc::kCondDim, 2 * D};
      h->specs[b + "." + nc + ".conditional_linear_layer.bias"] = {2 * D};
    }
  }
  h->specs[t + ".final_norm_cond.conditional_linear_layer.kernel"] = {gc::kCondDim, 2 * D};
  h->specs[t + ".final_norm_cond.conditional_linear_layer.bias"] = {2 * D};
}

// kernel (in,out) -> transposed [out_pad][in_pad], using input rows [in_begin, in_begin+in_count)
std::vector<float> transpose_pad(const std::vector<float>& k, int n_in, int n_out, int in_begin,
                                 int in_count, int in_pad, int out_pad) {
  std::vector<float> t((size_t)out_pad * in_pad, 0.f);
  for (int i = 0; i < in_count; ++i)
    for (int o = 0; o < n_out; ++o) t[(size_t)o * in_pad + i] = k[(size_t)(in_begin + i) * n_out + o];
  (void)n_in;
  return t;
}

// IEEE half conversions on the host (round to nearest even), used to pre-split the weights.
uint16_t f32_to_f16_bits(float f) {
  uint32_t x;
  std::memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  const uint32_t mant = x & 0x7FFFFFu;
  const int exp = (int)((x >> 23) & 0xFF) - 127 + 15;
  if (((x >> 23) & 0xFF) == 0xFF) return (uint16_t)(sign | 0x7C00u | (mant ? 0x200u : 0));
  if (exp >= 31) return (uint16_t)(sign | 0x7BFFu);                 // clamp to the largest finite half
  if (exp <= 0) {
    if (exp < -10) return (uint16_t)sign;
    const uint32_t m = mant | 0x800000u;
    const int shift = 14 - exp;
    uint32_t h = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (h & 1))) ++h;
    return (uint16_t)(sign | h);
  }
  uint32_t h = ((uint32_t)exp << 10) | (mant >> 13);
  const uint32_t rem = mant & 0x1FFFu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) ++h;
  return (uint16_t)(sign | h);
}

float f16_bits_to_f32(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  const int exp = (h >> 10) & 0x1F;
  const uint32_t mant = h & 0x3FFu;
  float out;
  if (exp == 0) {
    out = std::ldexp((float)mant, -24);
  } else if (exp == 31) {
    out = mant ? NAN : INFINITY;
  } else {
    out = std::ldexp((float)(mant | 0x400u), exp - 25);
  }
  uint32_t bits;
  std::memcpy(&bits, &out, 4);
  bits |= sign;
  std::memcpy(&out, &bits, 4);
  return out;
}

// Row-major f32 [rows][k] (k % 32 == 0) -> S16: per row, k/32 groups of [32 hi halfs | 32 lo halfs];
// same 4 bytes per element, returned as a float-typed buffer.
std::vector<float> encode_s16(const std::vector<float>& m, int rows, int k) {
  std::vector<float> out((size_t)rows * k);
  uint16_t* o = reinterpret_cast<uint16_t*>(out.data());
  for (int r = 0; r < rows; ++r)
    for (int g = 0; g < k / 32; ++g)
      for (int i = 0; i < 32; ++i) {
        float x = m[(size_t)r * k + g * 32 + i];
        x = std::min(std::max(x, -65000.0f), 65000.0f);
        const uint16_t hi = f32_to_f16_bits(x);
        const uint16_t lo = f32_to_f16_bits((x - f16_bits_to_f32(hi)) * 2048.0f);
        o[((size_t)r * k + g * 32) * 2 + i] = hi;
        o[((size_t)r * k + g * 32) * 2 + 32 + i] = lo;
      }
  return out;
}

// Row-major f32 W^T [n][k] (n % 32 == 0, k % 16 == 0) -> WF16, the MFMA fragment order the
// weight-streaming GEMM loads with one coalesced 16-byte read per lane: for column tile ct = n/32
// and k step s = k/16, 1 KB of hi halfs then 1 KB of lo halfs; inside each, lane (k%16/8)*32 + n%32
// holds the 8 consecutive k values it feeds to v_mfma_f32_32x32x16_f16.
std::vector<float> encode_wf16(const std::vector<float>& m, int n, int k) {
  std::vector<float> out((size_t)n * k);
  uint16_t* o = reinterpret_cast<uint16_t*>(out.data());
  const size_t steps = (size_t)k / 16;
  for (int row = 0; row < n; ++row)
    for (int kk = 0; kk < k; ++kk) {
      float x = m[(size_t)row * k + kk];
      x = std::min(std::max(x, -65000.0f), 65000.0f);
      const uint16_t hi = f32_to_f16_bits(x);
      const uint16_t lo = f32_to_f16_bits((x - f16_bits_to_f32(hi)) * 2048.0f);
      const size_t frag = ((size_t)(row / 32) * steps + kk / 16) * 2;
      const int k16 = kk % 16;
      const size_t lane = (size_t)(k16 >> 3) * 32 + row % 32;
      o[(frag * 64 + lane) * 8 + (k16 & 7)] = hi;
      o[((frag + 1) * 64 + lane) * 8 + (k16 & 7)] = lo;
    }
  return out;
}

// Row-major f32 W^T [n][k] -> WF32, the same blocks as WF16 with float32 payload: for column tile n/32 and k step k/16,
// 512 floats; lane (k%16/8)*32 + n%32 holds ITS 8 consecutive k values as 4 floats at [lane*4] (k%8 < 4) and 4 floats at
// [256 + lane*4] -- so the weight-streaming kernels' two 16-byte loads per fragment (the "hi" and "lo" slots of the
// ring) fetch the two halves, and a k16 step is 8 v_mfma_f32_32x32x2_f32 (exact-f32 family, precision = f32).
std::vector<float> encode_wf32(const std::vector<float>& m, int n, int k) {
  std::vector<float> out((size_t)n * k);
  const size_t steps = (size_t)k / 16;
  for (int row = 0; row < n; ++row)
    for (int kk = 0; kk < k; ++kk) {
      const size_t blk = (size_t)(row / 32) * steps + kk / 16;
      const size_t lane = (size_t)((kk % 16) / 8) * 32 + row % 32;
      out[blk * 512 + ((kk % 8) / 4) * 256 + lane * 4 + kk % 4] = m[(size_t)row * k + kk];
    }
  return out;
}

std::vector<float> pad_vec(const std::vector<float>& v, int n_pad) {
  std::vector<float> r(n_pad, 0.f);
  std::copy(v.begin(), v.end(), r.begin());
  return r;
}

struct CondPacker {
  std::vector<const std::vector<float>*> kernels, biases;
  std::vector<int> sizes;
  std::map<std::string, std::pair<int, int>> sites;   // parameter path of the conditioning linear -> (offset, width)
  int total = 0;
  int add(const std::string& name, const std::vector<float>& k, const std::vector<float>& b, int c) {
    kernels.push_back(&k); biases.push_back(&b); sizes.push_back(c);
    const int off = total;
    sites[name] = {off, c};
    total += 2 * c;
    return off;
  }
};

// one fused launch's weights: (k1 [n_in][n_hid], b1) -> activation -> (k2 [n_hid][n_out], b2)
int upload_mlp_pair(gc_handle* h, const std::vector<float>& k1, const std::vector<float>& b1,
                    const std::vector<float>& k2, const std::vector<float>& b2, int n_in, int in_begin, int in_count,
                    int in_pad, int n_hid, int n_out, DevMlp* out) {
  const int n_out_pad = round_up(n_out, 128);
  int rc;
  {
    const auto w1 = transpose_pad(k1, n_in, n_hid, in_begin, in_count, in_pad, n_hid);
    const auto w2 = transpose_pad(k2, n_hid, n_out, 0, n_hid, n_hid, n_out_pad);
    if ((rc = dev_upload(h, &out->w1t, w1))) return rc;
    if ((rc = dev_upload(h, &out->w2t, w2))) return rc;
    if ((rc = dev_upload(h, &out->w1s, encode_s16(w1, n_hid, in_pad)))) return rc;
    if ((rc = dev_upload(h, &out->w2s, encode_s16(w2, n_out_pad, n_hid)))) return rc;
    // weight-streaming images: K zero-padded to a multiple of 64 (the ring walks 4 k16 steps)
    out->k1f = round_up(in_pad, 64);
    const auto w1p = transpose_pad(k1, n_in, n_hid, in_begin, in_count, out->k1f, n_hid);
    if ((rc = dev_upload(h, &out->w1f, encode_wf16(w1p, n_hid, out->k1f)))) return rc;
    if ((rc = dev_upload(h, &out->w2f, encode_wf16(w2, n_out_pad, n_hid)))) return rc;
    if (h->f32_ws) {
      if ((rc = dev_upload(h, &out->w1x, encode_wf32(w1p, n_hid, out->k1f)))) return rc;
      if ((rc = dev_upload(h, &out->w2x, encode_wf32(w2, n_out_pad, n_hid)))) return rc;
    }
  }
  if ((rc = dev_upload(h, &out->b1, b1))) return rc;
  if ((rc = dev_upload(h, &out->b2, pad_vec(b2, n_out_pad)))) return rc;
  out->ldw1 = in_pad;
  out->n_out = n_out;
  out->n_out_pad = n_out_pad;
  out->cond_off = -1;
  return GC_OK;
}

int upload_mlp(gc_handle* h, const std::string& p, int n_in, int in_begin, int in_count, int in_pad,
               int n_hid, int n_out, bool cond, CondPacker* cp, DevMlp* out) {
  const int nh = h->hidden_layers;
  auto kern = [&](int i) -> const std::vector<float>& { return h->weights.at(p + ".network.network.layers." + std::to_string(2 * i) + ".kernel"); };
  auto bias = [&](int i) -> const std::vector<float>& { return h->weights.at(p + ".network.network.layers." + std::to_string(2 * i) + ".bias"); };
  int rc;
  out->pre.clear();
  if (nh == 1) {
    if ((rc = upload_mlp_pair(h, kern(0), bias(0), kern(1), bias(1), n_in, in_begin, in_count, in_pad, n_hid, n_out, out))) return rc;
  } else {
    // layers 0 .. nh-2: Linear -> activation, each as a fused launch whose second layer is the identity (no LayerNorm,
    // no conditioning): u = act(x W_i + b_i), u I + 0 = u.  Layer nh-1 and the output Linear are the usual pair.
    std::vector<float> eye((size_t)n_hid * n_hid, 0.f), zero((size_t)n_hid, 0.f);
    for (int i = 0; i < n_hid; ++i) eye[(size_t)i * n_hid + i] = 1.f;
    out->pre.resize(nh - 1);
    for (int i = 0; i + 1 < nh; ++i) {
      if (i == 0) rc = upload_mlp_pair(h, kern(0), bias(0), eye, zero, n_in, in_begin, in_count, in_pad, n_hid, n_hid, &out->pre[0]);
      else rc = upload_mlp_pair(h, kern(i), bias(i), eye, zero, n_hid, 0, n_hid, n_hid, n_hid, n_hid, &out->pre[i]);
      if (rc) return rc;
    }
    if ((rc = upload_mlp_pair(h, kern(nh - 1), bias(nh - 1), kern(nh), bias(nh), n_hid, 0, n_hid, n_hid, n_hid, n_out, out))) return rc;
  }
  if (cond) {
    const std::string c = p + ".norm_conditioning_layer.conditional_linear_layer";
    out->cond_off = cp->add(c, h->weights.at(c + ".kernel"), h->weights.at(c + ".bias"), n_out);
  }
  return GC_OK;
}

// ---- launch wrapper with optional per-class event bracketing --------------------------------
template <typename F>
int launch(gc_handle* h, int cls, F&& f) {
  ++h->launch_count;
  bool prof = (h->prof_cls == cls) && (h->prof_used + 2 <= h->prof_events.size());
  if (prof && h->prof_stride > 1) prof = ((h->prof_seen++ % (unsigned)h->prof_stride) == 0);
  if (prof) GC_HIP(h, hipEventRecord(h->prof_events[h->prof_used], h->stream));
  hipError_t e = f();
  if (e != hipSuccess) {
    h->err = std::string("launch ") + gc::kernel_class_name(cls) + ": " + hipGetErrorString(e);
    return GC_ERR_HIP;
  }
  if (prof) {
    GC_HIP(h, hipEventRecord(h->prof_events[h->prof_used + 1], h->stream));
    h->prof_used += 2;
  }
  return GC_OK;
}

// precision actually used by the next launches: f16x3 unless switched off, unsafe, or re-running
bool use_f16(const gc_handle* h) { return h->f16x3 && !h->weights_f16_unsafe && !h->in_fallback; }

gc::Segment seg(const float* ptr, const int* index, const float* affine, int width, int ld, int bcast) {
  gc::Segment s;
  s.ptr = ptr; s.index = index; s.affine = affine; s.width = width; s.ld = ld; s.bcast = bcast;
  return s;
}

// Arguments of one fused-MLP launch (run_mlp_one launches them; forward() also pairs two of them in one launch).
int build_mlp_args(gc_handle* h, const DevMlp& w, std::initializer_list<gc::Segment> segs, int rows, int B,
                   bool ln, bool cond, const float* residual, float* out, int ldo,
                   const gc::AddTerm* add0, const gc::AddTerm* add1, bool round_out, bool seg0_f32, bool out_f32,
                   bool tri, gc::MlpArgs* out_args) {
  gc::MlpArgs& a = *out_args;
  a = gc::MlpArgs{};
  a.nseg = 0;
  for (const auto& s : segs) a.seg[a.nseg++] = s;
  a.nadd = 0;
  if (add0) a.add[a.nadd++] = *add0;
  if (add1) a.add[a.nadd++] = *add1;
  a.rows = rows; a.B = B; a.hidden = h->cfg.latent_size;
  a.f16 = use_f16(h) ? 1 : 0;
  a.w1t = a.f16 ? w.w1s : w.w1t; a.ldw1 = w.ldw1; a.b1 = w.b1; a.w2t = a.f16 ? w.w2s : w.w2t; a.b2 = w.b2;
  const int k1e = w.k1e ? w.k1e : round_up(h->cfg.latent_size, 64);
  if (a.nadd) {   // split edge MLP: only the edge block of W1 multiplies the staged input (or, grid embedding: the noisy block)
    a.w1t = a.f16 ? w.w1e_s : w.w1e_t;
    a.ldw1 = w.ldw1e ? w.ldw1e : h->cfg.latent_size;
  }
  if (a.f16 && h->mlp_ws) {
    a.w1f = a.nadd ? w.w1e_f : w.w1f; a.k1f = a.nadd ? k1e : w.k1f;
    a.w2f = w.w2f; a.ones = h->d_ones; a.zeros = h->d_zeros;
  } else if (!a.f16 && h->mlp_ws && h->f32_ws && w.w1x && (!a.nadd || w.w1e_x)) {
    // exact-f32 family on the same weight-streaming kernel (WF32 images, v_mfma_f32_32x32x2_f32)
    a.w1f = a.nadd ? w.w1e_x : w.w1x; a.k1f = a.nadd ? k1e : w.k1f;
    a.w2f = w.w2x; a.ones = h->d_ones; a.zeros = h->d_zeros; a.f32w = 1;
  }
  a.n_out = w.n_out; a.n_out_pad = w.n_out_pad; a.do_ln = ln ? 1 : 0;
  a.cond = (cond && w.cond_off >= 0) ? (h->cond_cur ? h->cond_cur : h->d_cond) + w.cond_off : nullptr;
  a.cond_stride = h->cond_total;
  a.residual = residual; a.out = out; a.ldo = ldo;
  a.round16 = h->feat16 ? 1 : 0;
  a.round_out = (h->feat16 && round_out) ? 1 : 0;
  a.wt = (h->wt_stores & 2) ? 1 : 0;
  a.a16 = h->st16 ? 1 : 0;       // physical fp16 storage: the gc_a16 build (halfs in HBM, 2 MFMAs per product)
  a.seg0_f32 = seg0_f32 ? 1 : 0;
  a.out_f32 = out_f32 ? 1 : 0;
  a.tri = tri ? 1 : 0;
  if (tri && !gc::mlp_runs_weight_streaming(a))
    return fail(h, GC_ERR_INTERNAL, "the triple-sum epilogue exists in the weight-streaming MLP kernel only");
  return GC_OK;
}

int run_mlp_one(gc_handle* h, const DevMlp& w, std::initializer_list<gc::Segment> segs, int rows, int B,
                bool ln, bool cond, const float* residual, float* out, int ldo,
                const gc::AddTerm* add0 = nullptr, const gc::AddTerm* add1 = nullptr, bool round_out = true,
                hipStream_t on_stream = nullptr, bool seg0_f32 = false, bool out_f32 = false, bool tri = false) {
  gc::MlpArgs a{};
  if (int rc = build_mlp_args(h, w, segs, rows, B, ln, cond, residual, out, ldo, add0, add1, round_out, seg0_f32, out_f32, tri, &a))
    return rc;
  if (on_stream) {               // side stream: not bracketed by the per-class profiler (its events live on h->stream)
    ++h->launch_count;
    hipError_t e = a.a16 ? gc_a16::launch_mlp(on_stream, a16_view<gc_a16::MlpArgs>(a)) : gc::launch_mlp(on_stream, a);
    if (e != hipSuccess) return fail(h, GC_ERR_HIP, std::string("launch gc_mlp (side stream): ") + hipGetErrorString(e));
    return GC_OK;
  }
  return launch(h, gc::KC_MLP, [&] {
    return a.a16 ? gc_a16::launch_mlp(h->stream, a16_view<gc_a16::MlpArgs>(a)) : gc::launch_mlp(h->stream, a);
  });
}

// One MLPWithNormConditioning / MLP of the GNNs.  hidden_layers == 1 (the reference's trained configuration): one
// fused launch.  hidden_layers >= 2: the leading (Linear -> activation) layers run first, one launch each, handing a
// [rows][latent] float32 array over (gathers / concatenation happen in the first launch only).
int run_mlp(gc_handle* h, const DevMlp& w, std::initializer_list<gc::Segment> segs, int rows, int B,
            bool ln, bool cond, const float* residual, float* out, int ldo,
            const gc::AddTerm* add0 = nullptr, const gc::AddTerm* add1 = nullptr, bool round_out = true,
            hipStream_t on_stream = nullptr, bool seg0_f32 = false, bool out_f32 = false, bool tri = false) {
  if (w.pre.empty())
    return run_mlp_one(h, w, segs, rows, B, ln, cond, residual, out, ldo, add0, add1, round_out, on_stream, seg0_f32, out_f32, tri);
  if (add0 || add1 || on_stream || tri || !h->d_mlp_tmp[0]) return fail(h, GC_ERR_INTERNAL, "hidden_layers >= 2: unsupported MLP form");
  const int L = h->cfg.latent_size;
  int rc;
  if ((rc = run_mlp_one(h, w.pre[0], segs, rows, B, false, false, nullptr, h->d_mlp_tmp[0], L, nullptr, nullptr, false,
                        nullptr, seg0_f32, true)))
    return rc;
  int cur = 0;
  for (size_t i = 1; i < w.pre.size(); ++i) {
    if ((rc = run_mlp_one(h, w.pre[i], {seg(h->d_mlp_tmp[cur], nullptr, nullptr, L, L, 0)}, rows, B, false, false, nullptr,
                          h->d_mlp_tmp[cur ^ 1], L, nullptr, nullptr, false, nullptr, true, true)))
      return rc;
    cur ^= 1;
  }
  return run_mlp_one(h, w, {seg(h->d_mlp_tmp[cur], nullptr, nullptr, L, L, 0)}, rows, B, ln, cond, residual, out, ldo,
                     nullptr, nullptr, round_out, nullptr, true, out_f32);
}

// Row-tile height of the weight-streaming GEMM (x 32 rows).  Every workgroup streams its 128 weight columns
// once per row tile, so the L2 -> CU weight traffic per FLOP halves with every doubling: 64-row tiles once
// they still give >= 1.5 workgroups per CU.  128-row tiles (GC_TUNE_WS_MT=4) were measured at the 1-degree
// sizes and do not help (FFW-1 1.49 vs 1.42 ms per call): these GEMMs are not bound by weight traffic.
int pick_ws_mt(const gc_handle* h, int rows, int n, int splits) {
  if (h->ws_mt > 0) return h->ws_mt;
  const int panels = (n / 128) * splits;
  return ((rows + 63) / 64) * panels >= 400 ? 2 : 1;
}

// "fp16 node features" with PHYSICAL 2-byte activation storage (BASELINE.json configs[4]): every kernel of the call
// must then be the gc_a16 / H16 form that reads and writes halfs, so it is all or nothing -- when a shape or an A/B
// switch would put an LDS-staged or f32-MFMA kernel on the path (and during the exact-f32 re-run of the f16x3
// domain guard) the mode runs on float32 containers with the rounding flags instead (same values).
bool store16_ok(const gc_handle* h) {
  const gc_config& c = h->cfg;
  const int D = c.d_model, F = c.ffw_hidden;
  if (!(h->feat16 && h->a16 && use_f16(h) && h->gemm_ws && h->mlp_ws && h->attn_f16 && h->attn_v2 &&
        h->fuse_outrow && !h->side_stream))
    return false;
  if (D % 128 || D > 512) return false;
  if (h->attn_splits > 1 && !(h->attn_splits <= 8 && h->mt_out == 1 && h->fuse_combine)) return false;
  if (h->ffw_fused_slabs > 0) return true;
  return F % 128 == 0 && (F / h->ffw2_splits) % 128 == 0;     // both FFW layers as weight-streaming GEMMs
}

// Work-item list for an attention launch whose tiles (one workgroup each, one workgroup per CU at heads of 128) would
// run as a full round of 256 plus a partly filled second one that lasts nearly as long (321 tiles at the 1-degree size:
// list scheduling of its 12-14 chunks per tile gives 29.6 chunk-times against a balanced 18.1, tools/attention_tile_schedule.py).
// Every XCD takes a contiguous range of tiles (the L2 locality of the plain launch); 32 of them run whole, one per CU, the
// other n - 32 (evenly spaced inside the range) are cut into 32 key-range pieces of at most kItemPieces per tile that follow
// as a second, short round; the pieces' partial results are merged by the out-projection's loader (GemmArgs::att_tiles).
// items: [8 * per_xcd][4] = (tile, first chunk, end chunk, partial slot or -1), tile -1 = padding; tiles: [n_tiles][2] = (first
// slot, pieces).  Used where the tiles per XCD are ONE full round plus at most 16 (1 degree: 32 + 8 / 9).
bool build_attention_items(const gc::HostGraph& g, std::vector<int>* items, std::vector<int>* tiles) {
  const int n = g.n_tiles;
  tiles->assign((size_t)2 * n, 0);
  // per XCD: full rounds of whole tiles (32 CUs, one workgroup each), then the n_cut < 32 tiles left over as pieces.
  // Worth it while the pieces are shorter than the tiles they replace: at most 16 cut tiles (>= 2 pieces each).
  std::vector<std::vector<int>> lists(8);
  int slot = 0, any_cut = 0;
  for (int x = 0; x < 8; ++x) {
    const int t0 = (int)((long long)n * x / 8), t1 = (int)((long long)n * (x + 1) / 8), ng = t1 - t0;
    if (ng < 32) return false;                               // less than one round: the plain launch (with key splits) is the right one
    if (ng >= 64) return false;   // several rounds balance themselves: at 0.25 degree (5 rounds + 1 tile) the list measured 4.48 -> 4.40 ms of
                                  // attention per call and 2.57 -> 2.66 of out-projection, net nothing
    const int n_cut = ng % 32;
    if (n_cut > 16) return false;
    any_cut += n_cut;
    std::vector<char> cut(ng, 0);
    for (int k = 0; k < n_cut; ++k) cut[(int)(((2LL * k + 1) * ng) / (2LL * n_cut))] = 1;   // evenly spaced: distinct, ng / n_cut >= 2
    std::vector<int>& it = lists[x];
    std::vector<int> order;
    for (int i = 0; i < ng; ++i) {
      const int t = t0 + i;
      if (cut[i]) { order.push_back(t); continue; }
      it.insert(it.end(), {t, g.tile_chunk_start[t], g.tile_chunk_start[t + 1], -1});
    }
    if ((int)order.size() != n_cut) return false;
    if (n_cut == 0) continue;
    // up to 32 pieces (one short round) over the cut tiles, at most kItemPieces each; the longest tiles get the extra ones
    const int pieces = std::min(32, gc::kItemPieces * n_cut), base = pieces / n_cut, rem = pieces % n_cut;
    std::vector<int> by_len = order;
    std::stable_sort(by_len.begin(), by_len.end(), [&](int a, int b) {
      return g.tile_chunk_start[a + 1] - g.tile_chunk_start[a] > g.tile_chunk_start[b + 1] - g.tile_chunk_start[b];
    });
    std::vector<std::array<int, 4>> pieces_x;
    for (int t : order) {
      int np = base;
      for (int k = 0; k < rem; ++k)
        if (by_len[k] == t) ++np;
      const int c0 = g.tile_chunk_start[t], nc = g.tile_chunk_start[t + 1] - c0;
      np = std::min(np, std::max(nc, 1));                    // never more pieces than chunks
      if (np > gc::kItemPieces || np < 1) return false;
      (*tiles)[2 * t] = slot;
      (*tiles)[2 * t + 1] = np;
      for (int k = 0; k < np; ++k) pieces_x.push_back({t, c0 + (nc * k) / np, c0 + (nc * (k + 1)) / np, slot++});
    }
    // the second round is dealt to CUs as they finish their whole tile: longest piece first (list scheduling), so that the
    // last CUs to come free take the shortest pieces (GC_TUNE_ATTN_LPT=0: tile order)
    static const bool lpt = [] { const char* v = std::getenv("GC_TUNE_ATTN_LPT"); return !(v && *v == '0'); }();
    if (lpt)
      std::stable_sort(pieces_x.begin(), pieces_x.end(), [](const std::array<int, 4>& a, const std::array<int, 4>& b) {
        return a[2] - a[1] > b[2] - b[1];
      });
    for (const auto& pc : pieces_x) it.insert(it.end(), pc.begin(), pc.end());
  }
  if (any_cut == 0 || slot > n) return false;                // nothing to balance / the partial buffers hold n_tiles slots
  size_t per_xcd = 0;
  for (const auto& l : lists) per_xcd = std::max(per_xcd, l.size() / 4);
  items->assign(8 * per_xcd * 4, -1);                        // tile -1: padding
  for (int x = 0; x < 8; ++x) std::copy(lists[x].begin(), lists[x].end(), items->begin() + (size_t)x * per_xcd * 4);
  return true;
}

void drop_sample_graphs(gc_handle* h);

// Split first layer of the grid embedding for the current weights and noisy slots (gc_handle::embed_cache).
int build_embed_cache(gc_handle* h) {
  h->embed_cache_ready = false;
  if (!h->embed_cache || !h->finalized_weights || !h->has_slots || h->hidden_layers != 1 || !h->mlp_ws) return GC_OK;
  const gc_config& c = h->cfg;
  const int L = c.latent_size, node_in = 3 + c.c_in, kp = h->kp;
  const std::string p = std::string(P_G2M) + ".embedder_network.embed_node_fns.grid_nodes.network.network.layers.0.kernel";
  const auto& k1 = h->weights.at(p);                          // [node_in][L]
  int rc;
  drop_sample_graphs(h);                                      // captured samples bake these images' addresses
  auto wst = transpose_pad(k1, node_in, L, 0, node_in, kp, L);   // [L][kp]
  for (int o = 0; o < L; ++o)
    for (int cc = 0; cc < c.c_out; ++cc) wst[(size_t)o * kp + 3 + h->h_slots[cc]] = 0.f;
  if ((rc = dev_upload(h, &h->w1st_t, wst))) return rc;
  if ((rc = dev_upload(h, &h->w1st_s, encode_s16(wst, L, kp)))) return rc;
  h->nwp = round_up(c.c_out, 32);
  const int k1e = round_up(h->nwp, 64);
  std::vector<float> wn((size_t)L * k1e, 0.f);                // [L][k1e]: column cc = kernel row 3 + slots[cc]
  for (int cc = 0; cc < c.c_out; ++cc)
    for (int o = 0; o < L; ++o) wn[(size_t)o * k1e + cc] = k1[(size_t)(3 + h->h_slots[cc]) * L + o];
  DevMlp& n = h->g2m_embed_grid_n;
  n = h->g2m_embed_grid;                                      // second layer, biases, conditioning: shared
  n.w1e_t = n.w1e_s = nullptr;                                // (weight-streaming route only: embed_cache_usable)
  n.k1e = k1e;
  n.ldw1e = h->nwp;
  if ((rc = dev_upload(h, &n.w1e_f, encode_wf16(wn, L, k1e)))) return rc;
  n.w1e_x = nullptr;
  if (h->f32_ws && (rc = dev_upload(h, &n.w1e_x, encode_wf32(wn, L, k1e)))) return rc;
  if (!h->d_xn) {
    const size_t GB = (size_t)h->hg.G * c.batch;
    if ((rc = dev_alloc(h, &h->d_xn, GB * h->nwp))) return rc;
    if ((rc = dev_alloc(h, &h->d_pstat, GB * L))) return rc;
    GC_HIP(h, hipMemset(h->d_xn, 0, GB * h->nwp * sizeof(float)));   // the padding columns stay zero for good
  }
  h->embed_cache_ready = true;
  return GC_OK;
}

// May the sample about to be enqueued run on the cache?  (what run_mlp_one needs to put the noisy block on the
// weight-streaming kernel; fp16 node features round the staged inputs and keep the one-launch form.)
bool embed_cache_usable(const gc_handle* h) {
  if (!h->embed_cache_ready || h->feat16 || !h->mlp_ws || h->hidden_layers != 1) return false;
  if (h->cfg.latent_size == 512 && !h->mlp_ws512) return false;
  if (use_f16(h)) return true;
  return h->f32_ws && h->g2m_embed_grid_n.w1e_x != nullptr;
}

// mesh2grid edge update f1 = MLPc([f0 | m2[senders] | g1[receivers]]) (typed_graph_net.py:134-159,295-305; the first
// layer split by input block when h->split_edge: the per-node products are in d_pm / d_pg by then).  fused: the fused
// kernel's epilogue adds each grid node's three results and writes agg2 [G, L] instead of f1 [E2, L].
int run_m2g_edge(gc_handle* h, const float* cond, bool fused) {
  const gc::HostGraph& g = h->hg;
  const int B = h->cfg.batch, L = h->cfg.latent_size;
  float* out = fused ? h->d_agg2 : h->d_f1;
  if (h->split_edge) {
    const gc::AddTerm ts{h->d_pm, h->d_m2g_snd}, tr{h->d_pg, h->d_m2g_rcv};
    return run_mlp(h, h->m2g_edge, {seg(h->d_f0_hat, nullptr, cond + h->m2g_embed_edge.cond_off, L, L, 1)},
                   g.E2 * B, B, true, true, nullptr, out, L, &ts, &tr, true, nullptr, /*seg0_f32=*/true, false, fused);
  }
  return run_mlp(h, h->m2g_edge,
                 {seg(h->d_f0_hat, nullptr, cond + h->m2g_embed_edge.cond_off, L, L, 1),
                  seg(h->d_m2, h->d_m2g_snd, nullptr, L, L, 0),
                  seg(h->d_g1, h->d_m2g_rcv, nullptr, L, L, 0)},
                 g.E2 * B, B, true, true, nullptr, out, L, nullptr, nullptr, true, nullptr, /*seg0_f32=*/true, false, fused);
}

// the conditions under which run_mlp_one puts the mesh2grid edge MLP on the weight-streaming kernel (gc::launch_mlp)
bool m2g_sum_fusable(const gc_handle* h) {
  const DevMlp& w = h->m2g_edge;
  if (!h->m2g_fuse_sum || !h->hg.m2g_tri || !h->mlp_ws || !w.pre.empty()) return false;
  if (h->cfg.latent_size == 512 && !h->mlp_ws512) return false;
  if (use_f16(h)) return true;
  return h->f32_ws && w.w1x && (!h->split_edge || w.w1e_x);
}

// One denoiser forward on device-resident, already packed grid input (h->d_xp).
// sigma comes from h->d_sigma when sigma_scalar < 0, else the scalar is used for every batch element.
int forward(gc_handle* h, float sigma_scalar, const float* cond_ready = nullptr) {
  const gc_config& c = h->cfg;
  const gc::HostGraph& g = h->hg;
  const int B = c.batch, L = c.latent_size, D = c.d_model, F = c.ffw_hidden;
  hipStream_t s = h->stream;
  int rc;
  // cond_ready: the sampler computed this call's conditioning vectors up front (one launch per sample)
  const float* cond = cond_ready ? cond_ready : h->d_cond;
  h->cond_cur = cond;
  const int cs = h->cond_total;
  const int64_t launches0 = h->launch_count;
  const bool st16 = store16_ok(h);
  h->st16 = h->last_st16 = st16;

  if (!cond_ready && (rc = launch(h, gc::KC_COND, [&] {
         return gc::launch_cond(s, sigma_scalar < 0 ? h->d_sigma : nullptr, sigma_scalar, B, h->d_nw0t,
                                h->d_nb0, h->d_nw1t, h->d_nb1, c.noise_num_frequencies, c.noise_hidden,
                                c.noise_base_period, h->d_wc_all, h->d_bc_all, cs, h->d_condvec,
                                h->d_cond);
       })))
    return rc;

  // ---- grid2mesh (denoiser.py:602-688; deep_typed_graph_net.py:493-581) ----
  if (h->embed_cache_live) {   // inside a sample: the noisy block of the first layer + the cached static part (embed_cache)
    const gc::AddTerm ps{h->d_pstat, nullptr};
    if ((rc = run_mlp(h, h->g2m_embed_grid_n, {seg(h->d_xn, nullptr, nullptr, h->nwp, h->nwp, 0)}, g.G * B, B,
                      true, true, nullptr, h->d_g0, L, &ps, nullptr, true, nullptr, /*seg0_f32=*/true)))
      return rc;
  } else if ((rc = run_mlp(h, h->g2m_embed_grid, {seg(h->d_xp, nullptr, nullptr, h->kp, h->kp, 0)}, g.G * B, B,
                    true, true, nullptr, h->d_g0, L, nullptr, nullptr, true, nullptr, /*seg0_f32=*/true)))
    return rc;
  if ((rc = launch(h, gc::KC_PACK, [&] {
         return gc::launch_affine_rows(s, h->d_m0_hat, cond + h->g2m_embed_mesh.cond_off, cs, g.M, B, L,
                                       h->d_m0, h->feat16, st16);
       })))
    return rc;
  // per-node halves of an edge MLP's first layer: out[rows][L] = nodes[rows][L] @ W_block
  auto node_gemm = [&](const float* nodes, int rows, const float* wt_f32, const float* wt_s16, const float* wt_f16, float* out) {
    gc::GemmArgs ga{};
    ga.a = nodes; ga.lda = L; ga.a_f32 = 1; ga.ldw = L;
    ga.rows = rows; ga.n = L; ga.k_slice = L; ga.bias = nullptr; ga.act = 0; ga.out = out; ga.ldo = L;
    ga.round16 = 0;   // pre-activation terms of the split edge MLP: accumulator values, never rounded
    ga.out_f32 = 1;   // ... and float32 also when the node latents they are made from are stored as halfs
    if (use_f16(h) && h->gemm_ws && L % 128 == 0) {          // weight-streaming GEMM (nodes are halfs in the gc_a16 build)
      ga.wt = wt_f16; ga.a16 = st16 ? 1 : 0;
      const int ws_mt = pick_ws_mt(h, rows, L, 1);
      return launch(h, gc::KC_GEMM_NODE, [&] {
        return ga.a16 ? gc_a16::launch_gemm_ws(s, gc::KC_GEMM_NODE, a16_view<gc_a16::GemmArgs>(ga), ws_mt, 1, 0)
                      : gc::launch_gemm_ws(s, gc::KC_GEMM_NODE, ga, ws_mt, 1, 0);
      });
    }
    ga.wt = use_f16(h) ? wt_s16 : wt_f32;
    return launch(h, gc::KC_GEMM_NODE,
                  [&] { return gc::launch_gemm(s, gc::KC_GEMM_NODE, ga, 1, 1, 0, use_f16(h)); });
  };
  // The grid2mesh edge update and the grid-node update g1 = g0 + MLP(g0) read only g0 / m0 and are independent
  // (typed_graph_net.py:134-195): when both take the same kernel form they go out as ONE launch (gc_mlp_ws_pair_kernel) --
  // at nano the edge update's 526 row tiles are a round of 512 workgroups and a round of 14 that lasts as long again, and
  // the node update's 329 tiles fill that second round instead of a launch of their own (GC_TUNE_MLP_PAIR=0: two launches).
  bool paired = false;
  if (h->mlp_pair && !h->side_stream && h->g2m_edge.pre.empty() && h->g2m_grid.pre.empty()) {
    gc::MlpArgs ea{}, na{};
    const gc::AddTerm ts{h->d_pg, h->d_g2m_snd}, tr{h->d_pm, h->d_g2m_rcv};
    if (h->split_edge)
      rc = build_mlp_args(h, h->g2m_edge, {seg(h->d_e0_hat, nullptr, cond + h->g2m_embed_edge.cond_off, L, L, 1)},
                          g.E1 * B, B, true, true, nullptr, h->d_e1, L, &ts, &tr, true, /*seg0_f32=*/true, false, false, &ea);
    else
      rc = build_mlp_args(h, h->g2m_edge,
                          {seg(h->d_e0_hat, nullptr, cond + h->g2m_embed_edge.cond_off, L, L, 1),
                           seg(h->d_g0, h->d_g2m_snd, nullptr, L, L, 0), seg(h->d_m0, h->d_g2m_rcv, nullptr, L, L, 0)},
                          g.E1 * B, B, true, true, nullptr, h->d_e1, L, nullptr, nullptr, true, /*seg0_f32=*/true, false, false, &ea);
    if (rc) return rc;
    if ((rc = build_mlp_args(h, h->g2m_grid, {seg(h->d_g0, nullptr, nullptr, L, L, 0)}, g.G * B, B, true, true, h->d_g0, h->d_g1, L,
                             nullptr, nullptr, true, false, false, false, &na)))
      return rc;
    if (gc::mlp_pair_supported(ea, na)) {
      if (h->split_edge) {                       // the per-node halves of the edge MLP's first layer come first, as ever
        if ((rc = node_gemm(h->d_g0, g.G * B, h->g2m_edge.w1snd_t, h->g2m_edge.w1snd_s, h->g2m_edge.w1snd_f, h->d_pg))) return rc;
        if ((rc = node_gemm(h->d_m0, g.M * B, h->g2m_edge.w1rcv_t, h->g2m_edge.w1rcv_s, h->g2m_edge.w1rcv_f, h->d_pm))) return rc;
      }
      if ((rc = launch(h, gc::KC_MLP, [&] {
             return ea.a16 ? gc_a16::launch_mlp_pair(s, a16_view<gc_a16::MlpArgs>(ea), a16_view<gc_a16::MlpArgs>(na))
                           : gc::launch_mlp_pair(s, ea, na);
           })))
        return rc;
      paired = true;
    }
  }
  if (paired) {
  } else if (h->split_edge) {
    if ((rc = node_gemm(h->d_g0, g.G * B, h->g2m_edge.w1snd_t, h->g2m_edge.w1snd_s, h->g2m_edge.w1snd_f, h->d_pg))) return rc;
    if ((rc = node_gemm(h->d_m0, g.M * B, h->g2m_edge.w1rcv_t, h->g2m_edge.w1rcv_s, h->g2m_edge.w1rcv_f, h->d_pm))) return rc;
    const gc::AddTerm ts{h->d_pg, h->d_g2m_snd}, tr{h->d_pm, h->d_g2m_rcv};
    if ((rc = run_mlp(h, h->g2m_edge,
                      {seg(h->d_e0_hat, nullptr, cond + h->g2m_embed_edge.cond_off, L, L, 1)},
                      g.E1 * B, B, true, true, nullptr, h->d_e1, L, &ts, &tr, true, nullptr, /*seg0_f32=*/true)))
      return rc;
  } else if ((rc = run_mlp(h, h->g2m_edge,
                    {seg(h->d_e0_hat, nullptr, cond + h->g2m_embed_edge.cond_off, L, L, 1),
                     seg(h->d_g0, h->d_g2m_snd, nullptr, L, L, 0),
                     seg(h->d_m0, h->d_g2m_rcv, nullptr, L, L, 0)},
                    g.E1 * B, B, true, true, nullptr, h->d_e1, L, nullptr, nullptr, true, nullptr, /*seg0_f32=*/true)))
    return rc;
  if ((rc = launch(h, gc::KC_SEGSUM, [&] {
         return gc::launch_segsum(s, h->d_e1, h->d_g2m_ptr, h->d_g2m_eid, g.M, g.E1, B, L, h->d_agg1, h->feat16, st16,
                                  h->g2m_agg_norm);
       })))
    return rc;
  if ((rc = run_mlp(h, h->g2m_mesh,
                    {seg(h->d_m0, nullptr, nullptr, L, L, 0), seg(h->d_agg1, nullptr, nullptr, L, L, 0)},
                    g.M * B, B, true, true, h->d_m0, h->d_x, L)))
    return rc;
  // The grid-node update g1 = g0 + MLP(g0) feeds nothing before the mesh2grid edge update, 1.1 ms
  // later, so it CAN run on a side stream beside the mesh path (GC_TUNE_SIDE_STREAM=1).  Measured: the two
  // cross-stream event waits cost more than the 19-us kernel hides (688 vs 706 calls/s) -- off by default.
  const bool side = h->side_stream && h->stream2 && h->prof_cls < 0;
  if (side) {
    GC_HIP(h, hipEventRecord(h->ev_fork, s));
    GC_HIP(h, hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
  }
  if (!paired && (rc = run_mlp(h, h->g2m_grid, {seg(h->d_g0, nullptr, nullptr, L, L, 0)}, g.G * B, B, true, true,
                               h->d_g0, h->d_g1, L, nullptr, nullptr, true, side ? h->stream2 : nullptr)))
    return rc;
  if (side) GC_HIP(h, hipEventRecord(h->ev_join, h->stream2));

  // ---- mesh transformer (sparse_transformer.py:486-525, 624-634) ----
  // The residual adds are deferred: a projection writes split-K slabs, and the next row pass
  // (gc_rowop) folds "x += bias + slabs" together with the following LayerNorm + conditioning.
  const int MB = g.M * B;
  const int n_layers = (h->debug_layer_limit >= 0 && h->debug_layer_limit < c.num_layers)
                           ? h->debug_layer_limit : c.num_layers;
  const float* pend_bias = nullptr;
  int pend_slabs = 0;
  const bool f16 = use_f16(h);
  // exact-f32 family (precision = f32, and the re-run of the f16x3 domain guard) on the SAME launch structure: the
  // weight-streaming GEMM, the fused FFW and the out-projection + row pass read WF32 images and multiply on
  // v_mfma_f32_32x32x2_f32 (round 4; before: LDS-staged GEMMs, no fused FFW, 123 launches per call)
  const bool x32 = !f16 && h->f32_ws && h->gemm_ws && !h->layers.empty() && h->layers[0].w1_x != nullptr;
  const int ffw_slabs = ((f16 || x32) && h->gemm_ws) ? h->ffw_fused_slabs : 0;   // precision can be switched after gc_finalize
  auto rowop = [&](const float* bias, int slabs, int cond_off, float* hout) {
    return launch(h, gc::KC_ROWOP, [&] {
      // (experiment) behind a fused FFW whose row tiles were pinned to XCDs, the row pass reads XCD-local slabs
      const int xr = (h->ffw_xcd && ffw_slabs > 0 && slabs == ffw_slabs && D == 256) ? 96 : 0;
      return gc::launch_rowop(s, h->d_x, bias, h->d_part, slabs, MB, D, B, cond + cond_off, cs, hout, 0, h->feat16, st16, xr);
    });
  };
  // f16x3: the weight-streaming kernel (WF16 weights) whenever the K slice is a multiple of 128
  auto use_ws = [&](int n, int k, int splits) {
    return (f16 || x32) && h->gemm_ws && n % 128 == 0 && (k / splits) % 128 == 0;
  };
  auto gemm = [&](int cls, const float* a, int lda, const float* wt, const float* wf, int ldw, int n, int k,
                  int splits, const float* bias, int act, float* out, int ldo, int mt, int epi) {
    gc::GemmArgs ga{};
    ga.a = a; ga.lda = lda; ga.a_f32 = 1; ga.ldw = ldw; ga.rows = MB; ga.n = n; ga.k_slice = k / splits;
    ga.bias = bias; ga.act = act; ga.out = out; ga.ldo = ldo; ga.round16 = h->feat16 ? 1 : 0; ga.a16 = st16 ? 1 : 0;
    if (use_ws(n, k, splits)) {
      // 64-row tiles halve the weight traffic; worth it once they still give >= 1.5 tiles per CU
      const int ws_mt = std::min(pick_ws_mt(h, MB, n, splits), x32 ? 2 : 4);
      ga.wt = wf;                              // (the caller passes the WF32 image in exact-f32 mode)
      ga.f32w = x32 ? 1 : 0;
      return launch(h, cls, [&] {
        return ga.a16 ? gc_a16::launch_gemm_ws(s, cls, a16_view<gc_a16::GemmArgs>(ga), ws_mt, splits, epi)
                      : gc::launch_gemm_ws(s, cls, ga, ws_mt, splits, epi);
      });
    }
    ga.wt = wt;
    return launch(h, cls, [&] { return gc::launch_gemm(s, cls, ga, mt, splits, epi, f16); });
  };
  // gc_debug_set_stop (tests): leave the forward inside block i, after phase 0 (pre-attention row pass: x, h),
  // 1 (QKV projection) or 2 (attention + out-projection + row pass: x, h); buffers keep what was computed so far
  auto stop_here = [&](int i, int phase) { return h->debug_stop_layer == i && h->debug_stop_phase == phase; };
  // attention as a work-item list (build_attention_items): needs the v2 kernel and the out-projection whose loader merges
  // (f16x3: the v2 kernel; exact f32: gc_attention_kernel takes the same list since round 5)
  const bool use_items = h->att_n_items > 0 && ((f16 && h->attn_f16 && h->attn_v2) || x32) && use_ws(3 * D, D, 1) &&
                         h->attn_splits == 1 && h->gemm_ws && h->fuse_outrow && D % 128 == 0 && D <= 512;
  h->last_att_items = use_items ? h->att_n_items : 0;
  for (int i = 0; i < n_layers; ++i) {
    const DevLayer& ly = h->layers[i];
    if ((rc = rowop(pend_bias, pend_slabs, ly.cond_attn, h->d_h))) return rc;
    if (stop_here(i, 0)) return GC_OK;
    // f16x3: the projection hands K and V to attention already split into fp16 hi / lo planes
    const bool v2 = f16 && h->attn_f16 && h->attn_v2 && use_ws(3 * D, D, 1);
    h->kv16_live = v2;
    if (v2) {
      gc::GemmArgs ga{};
      ga.a = h->d_h; ga.lda = D; ga.a_f32 = 1; ga.wt = ly.wqkv_f; ga.ldw = D; ga.rows = MB; ga.n = 3 * D; ga.k_slice = D;
      ga.out = h->d_qkv; ga.ldo = st16 ? D : 3 * D;   // fp16 storage: q alone, as halfs [rows][D]
      ga.round16 = h->feat16 ? 1 : 0; ga.a16 = st16 ? 1 : 0; ga.kv16 = h->d_kv16; ga.kv_d = D;
      const int ws_mt = pick_ws_mt(h, MB, 3 * D, 1);
      if ((rc = launch(h, gc::KC_GEMM_QKV, [&] {
             return ga.a16 ? gc_a16::launch_gemm_ws(s, gc::KC_GEMM_QKV, a16_view<gc_a16::GemmArgs>(ga), ws_mt, 1, 3)
                           : gc::launch_gemm_ws(s, gc::KC_GEMM_QKV, ga, ws_mt, 1, 3);
           })))
        return rc;
      if (stop_here(i, 1)) return GC_OK;
      if ((rc = launch(h, gc::KC_ATTN, [&] {
             return gc::launch_attention_v2(s, h->d_qkv, h->d_kv16, h->d_att, h->d_apart_o, h->d_apart_ml, g.M, B, D,
                                            c.num_heads, h->attn_splits, h->d_tile_start, h->d_union, h->d_mask,
                                            g.n_tiles, h->max_tile_chunks, h->feat16, st16,
                                            use_items ? h->d_att_items : nullptr, use_items ? h->att_n_items : 0);
           })))
        return rc;
    } else {
    if ((rc = gemm(gc::KC_GEMM_QKV, h->d_h, D, f16 ? ly.wqkv_s : ly.wqkv_t, x32 ? ly.wqkv_x : ly.wqkv_f, D, 3 * D, D, 1, nullptr, 0,
                   h->d_qkv, 3 * D, h->mt_qkv, 0)))
      return rc;
    if (stop_here(i, 1)) return GC_OK;
    if ((rc = launch(h, gc::KC_ATTN, [&] {
           return gc::launch_attention(s, h->d_qkv, h->d_att, h->d_apart_o, h->d_apart_ml, g.M, B, D,
                                       c.num_heads, h->attn_splits, false, h->d_tile_start, h->d_union,
                                       h->d_mask, g.n_tiles, f16 && h->attn_f16, h->max_tile_chunks, h->feat16,
                                       use_items ? h->d_att_items : nullptr, use_items ? h->att_n_items : 0);
         })))
      return rc;
    }
    // key-split partials are merged inside the out-projection's A loader (no combine launch) when
    // the projection runs with 32-row tiles and there are at most 4 splits
    const bool fuse_combine = h->attn_splits > 1 && h->attn_splits <= 8 && h->mt_out == 1 && h->fuse_combine;
    if (h->attn_splits > 1 && !fuse_combine && (rc = launch(h, gc::KC_ATTN_COMBINE, [&] {
          return gc::launch_attn_combine(s, h->d_apart_o, h->d_apart_ml, g.M, B, D, c.num_heads,
                                         h->attn_splits, h->d_att, false, h->feat16);
        })))
      return rc;
    // out-projection with the row pass in its epilogue (f16x3 weight-streaming form, no K split)
    const bool fuse_row = (f16 || x32) && h->gemm_ws && h->fuse_outrow && D % 128 == 0 && D <= 512 &&
                          (h->attn_splits == 1 || fuse_combine);
    if (fuse_row) {
      gc::GemmArgs ga{};
      ga.a = h->d_att; ga.lda = D; ga.a_f32 = 1; ga.wt = x32 ? ly.wo_x : ly.wo_f; ga.ldw = D; ga.rows = MB; ga.n = D; ga.k_slice = D;
      ga.f32w = x32 ? 1 : 0;
      if (h->attn_splits > 1) {
        ga.att_po = h->d_apart_o; ga.att_pml = h->d_apart_ml; ga.att_S = h->attn_splits; ga.att_B = B;
        ga.att_H = c.num_heads; ga.att_DH = D / c.num_heads;
      } else if (use_items) {                    // rows of the tiles the item list cut into pieces: merged in the loader
        ga.att_po = h->d_apart_o; ga.att_pml = h->d_apart_ml; ga.att_S = 0; ga.att_B = B;
        ga.att_H = c.num_heads; ga.att_DH = D / c.num_heads; ga.att_tiles = h->d_att_tiles;
      }
      ga.round16 = h->feat16 ? 1 : 0; ga.a16 = st16 ? 1 : 0;
      gc::RowFuse rf{h->d_x, ly.bo, cond + ly.cond_ffw, cs, B, h->d_h, h->feat16 ? 1 : 0};
      if ((rc = launch(h, gc::KC_GEMM_OUT, [&] {
             return ga.a16 ? gc_a16::launch_gemm_rowop(s, gc::KC_GEMM_OUT, a16_view<gc_a16::GemmArgs>(ga),
                                                       a16_view<gc_a16::RowFuse>(rf))
                           : gc::launch_gemm_rowop(s, gc::KC_GEMM_OUT, ga, rf);
           })))
        return rc;
    } else if (fuse_combine) {
      gc::GemmArgs ga{};
      const bool ws = use_ws(D, D, h->out_splits);
      ga.a = h->d_att; ga.lda = D; ga.a_f32 = 1; ga.wt = ws ? (x32 ? ly.wo_x : ly.wo_f) : (f16 ? ly.wo_s : ly.wo_t); ga.ldw = D; ga.rows = MB;
      ga.f32w = (ws && x32) ? 1 : 0;
      ga.n = D; ga.k_slice = D / h->out_splits; ga.out = h->d_part; ga.ldo = D;
      ga.att_po = h->d_apart_o; ga.att_pml = h->d_apart_ml; ga.att_S = h->attn_splits; ga.att_B = B;
      ga.att_H = c.num_heads; ga.att_DH = D / c.num_heads; ga.round16 = h->feat16 ? 1 : 0; ga.a16 = st16 ? 1 : 0;
      if ((rc = launch(h, gc::KC_GEMM_OUT, [&] {
             return ws ? (ga.a16 ? gc_a16::launch_gemm_ws(s, gc::KC_GEMM_OUT, a16_view<gc_a16::GemmArgs>(ga), 1, h->out_splits, 1)
                                 : gc::launch_gemm_ws(s, gc::KC_GEMM_OUT, ga, 1, h->out_splits, 1))
                       : gc::launch_gemm(s, gc::KC_GEMM_OUT, ga, 1, h->out_splits, 1, f16);
           })))
        return rc;
    } else if ((rc = gemm(gc::KC_GEMM_OUT, h->d_att, D, f16 ? ly.wo_s : ly.wo_t, x32 ? ly.wo_x : ly.wo_f, D, D, D, h->out_splits,
                          nullptr, 0, h->d_part, D, h->mt_out, 1)))
      return rc;
    if (!fuse_row && (rc = rowop(ly.bo, h->out_splits, ly.cond_ffw, h->d_h))) return rc;
    if (stop_here(i, 2)) return GC_OK;
    if (ffw_slabs > 0) {   // both FFW layers in one launch, one slab per 256 hidden columns
      gc::FfwArgs fa{h->d_h, MB, (int)D, (int)F, x32 ? ly.w1_x : ly.w1_f, ly.b1, x32 ? ly.w2_x : ly.w2_f, h->d_part,
                     h->feat16 ? 1 : 0, h->wt_stores & 1, st16 ? 1 : 0};
      fa.f32w = x32 ? 1 : 0;
      fa.xcd_tiles = (h->ffw_xcd && D == 256) ? 1 : 0;
      if ((rc = launch(h, gc::KC_GEMM_FFW1, [&] {
             return fa.a16 ? gc_a16::launch_ffw_fused(s, a16_view<gc_a16::FfwArgs>(fa)) : gc::launch_ffw_fused(s, fa);
           })))
        return rc;
    } else {
    if ((rc = gemm(gc::KC_GEMM_FFW1, h->d_h, D, f16 ? ly.w1_s : ly.w1_t, x32 ? ly.w1_x : ly.w1_f, D, F, D, 1, ly.b1, 1, h->d_u, F,
                   h->mt_ffw1, 0)))
      return rc;
    if ((rc = gemm(gc::KC_GEMM_FFW2, h->d_u, F, f16 ? ly.w2_s : ly.w2_t, x32 ? ly.w2_x : ly.w2_f, F, D, F, h->ffw2_splits, nullptr, 0,
                   h->d_part, D, h->mt_ffw2, 1)))
      return rc;
    }
    pend_bias = ly.b2;
    pend_slabs = ffw_slabs > 0 ? ffw_slabs : h->ffw2_splits;
  }
  if ((rc = rowop(pend_bias, pend_slabs, h->cond_final, h->d_m2))) return rc;

  if (side) GC_HIP(h, hipStreamWaitEvent(s, h->ev_join, 0));     // g1 is needed from here on
  // ---- mesh2grid + decoder (denoiser.py:730-768) ----
  if (h->split_edge) {
    if ((rc = node_gemm(h->d_m2, g.M * B, h->m2g_edge.w1snd_t, h->m2g_edge.w1snd_s, h->m2g_edge.w1snd_f, h->d_pm))) return rc;
    if ((rc = node_gemm(h->d_g1, g.G * B, h->m2g_edge.w1rcv_t, h->m2g_edge.w1rcv_s, h->m2g_edge.w1rcv_f, h->d_pg))) return rc;
  }
  // The edge update, and the sum of every grid node's 3 updated edges (typed_graph_net.py:175-182): ONE launch when the
  // edge set is the reference's (3 edges per grid node, kept sorted by receiver: HostGraph::m2g_tri) -- the sum happens
  // in the fused MLP's epilogue and f1 [E2, L] is neither stored nor read back.  Other in-degrees (injected graphs),
  // hidden_layers >= 2, the LDS-staged MLP kernel: edge update, then the segment-sum launch.
  const bool fuse_sum = m2g_sum_fusable(h);
  h->last_m2g_fused = fuse_sum;
  if ((rc = run_m2g_edge(h, cond, fuse_sum))) return rc;
  if (!fuse_sum && (rc = launch(h, gc::KC_SEGSUM, [&] {
         return gc::launch_segsum(s, h->d_f1, h->d_m2g_ptr, h->d_m2g_eid, g.G, g.E2, B, L, h->d_agg2, h->feat16, st16);
       })))
    return rc;
  if ((rc = run_mlp(h, h->m2g_grid,
                    {seg(h->d_g1, nullptr, nullptr, L, L, 0), seg(h->d_agg2, nullptr, nullptr, L, L, 0)},
                    g.G * B, B, true, true, h->d_g1, h->d_g2, L)))
    return rc;
  if ((rc = run_mlp(h, h->m2g_dec, {seg(h->d_g2, nullptr, nullptr, L, L, 0)}, g.G * B, B, false, false,
                    nullptr, h->d_y, c.c_out, nullptr, nullptr, true, nullptr, false, /*out_f32=*/true)))
    return rc;
  h->launches_last_call = h->launch_count - launches0;
  return GC_OK;
}

// ---- f16x3 domain guard -------------------------------------------------------------------------
// Enqueues the finite check of `p` and the copy of the counter to pinned host memory.
int guard_enqueue(gc_handle* h, const float* p, size_t n) {
  if (!use_f16(h)) return GC_OK;
  int rc = launch(h, gc::KC_PACK, [&] { return gc::launch_finite_check(h->stream, p, n, h->d_nonfinite); });
  if (rc) return rc;
  GC_HIP(h, hipMemcpyAsync(h->h_nonfinite, h->d_nonfinite, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
  return GC_OK;
}
// After the stream has been synchronised: did the check that guard_enqueue queued see NaN / Inf?
bool guard_tripped(gc_handle* h) {
  if (*h->h_nonfinite == h->nonfinite_seen) return false;
  h->nonfinite_seen = *h->h_nonfinite;
  return true;
}

// Static embeddings: LayerNorm(MLP(static features)) in the precision currently selected; the
// per-call conditioning is applied where they are consumed.  Re-run when the precision changes.
int compute_static_embeddings(gc_handle* h) {
  const gc::HostGraph& hg = h->hg;
  const int L = h->cfg.latent_size;
  int rc;
  h->st16 = store16_ok(h);       // same kernel build as the forward will use; float32 in (structural features) and out
  if ((rc = run_mlp(h, h->g2m_embed_mesh, {seg(h->d_mesh_struct16, nullptr, nullptr, 32, 32, 1)}, hg.M, 1, true, false, nullptr, h->d_m0_hat, L, nullptr, nullptr, false, nullptr, true, true))) return rc;
  if ((rc = run_mlp(h, h->g2m_embed_edge, {seg(h->d_e1_struct16, nullptr, nullptr, 32, 32, 1)}, hg.E1, 1, true, false, nullptr, h->d_e0_hat, L, nullptr, nullptr, false, nullptr, true, true))) return rc;
  if ((rc = run_mlp(h, h->m2g_embed_edge, {seg(h->d_e2_struct16, nullptr, nullptr, 32, 32, 1)}, hg.E2, 1, true, false, nullptr, h->d_f0_hat, L, nullptr, nullptr, false, nullptr, true, true))) return rc;
  GC_HIP(h, hipStreamSynchronize(h->stream));
  return GC_OK;
}

// f32 scalar arithmetic of the preconditioning (dpm_solver_plus_plus_2s.py:181-205, sigma_data = 1)
float f_c_in(float s) { return 1.0f / std::sqrt(s * s + 1.0f); }
float f_c_out(float s) { return s / std::sqrt(s * s + 1.0f); }
float f_c_skip(float s) { return 1.0f / (s * s + 1.0f); }

// out = (base ? base : 0) + scale * (a fresh unit-variance spherical white-noise field [G, B, c_out])
int noise_field(gc_handle* h, const float* base, float scale, float* out) {
  if (h->nz_L == 0) return fail(h, GC_ERR_STATE, "gc_noise_set_tables has not been called");
  const int N = h->cfg.batch * h->cfg.c_out;
  const size_t ncoef = (size_t)2 * h->nz_L * h->nz_L * N;
  const unsigned long long stream = h->nz_stream++;
  int rc = launch(h, gc::KC_NOISE, [&] { return gc::launch_noise_normals(h->stream, h->d_nz_coef, ncoef, h->nz_key, stream); });
  if (rc) return rc;
  return launch(h, gc::KC_NOISE, [&] {
    return gc::launch_noise_synthesis(h->stream, h->d_nz_leg, h->d_nz_cos, h->d_nz_sin, h->d_nz_coef, h->d_nz_f,
                                      h->nz_L, h->nz_lat, h->nz_lon, N, base, scale, out);
  });
}

// Everything one sample enqueues on h->stream between the two timing events: kernel launches and one device-to-device
// copy, no allocation, no host synchronisation -- so it can run under stream capture (run_sampler).
int sampler_body(gc_handle* h, const float* sigmas, int n, int skip_dead, const std::vector<float>& call_sigma,
                 bool multi, int* calls_out) {
  const gc_config& c = h->cfg;
  const int rows = h->hg.G * c.batch;
  const size_t ne = (size_t)rows * c.c_out;
  hipStream_t s = h->stream;
  int rc, calls = 0;
  const size_t cond_call = (size_t)c.batch * h->cond_total;
  const bool churn = !h->churn_rates.empty();
  // The kernel that produces a state also writes it, times the next call's c_in, into the noisy-target slots of the
  // packed grid input (one launch per denoiser call less); with stochastic churn in front of a call the state changes
  // once more first, and the write stays a launch of its own (GC_TUNE_FUSE_NOISY=0: always).
  // The per-sample-constant part of the grid embedding's first layer, once per sample (gc_handle::embed_cache): the
  // noisy columns of the packed input are cleared (their weights are zero in w1st, but 0 x NaN is NaN), then
  // P = xp @ W1_static (the LDS-staged GEMM: K = kp is a multiple of 32, not of 128; float32 out, no bias -- the
  // MLP adds b1 itself).  From here on the sampler writes the noisy channels into the compact array d_xn.
  const bool cache = embed_cache_usable(h);
  h->embed_cache_live = cache;
  struct CacheOff { gc_handle* h; ~CacheOff() { h->embed_cache_live = false; } } cache_off{h};   // gc_denoise keeps the full form
  if (cache) {
    if ((rc = launch(h, gc::KC_PACK, [&] { return gc::launch_zero_slots(s, h->d_slots, rows, c.c_out, h->kp, h->d_xp); }))) return rc;
    gc::GemmArgs ga{};
    ga.a = h->d_xp; ga.lda = h->kp; ga.a_f32 = 1; ga.ldw = h->kp; ga.rows = rows; ga.n = c.latent_size; ga.k_slice = h->kp;
    ga.bias = nullptr; ga.act = 0; ga.out = h->d_pstat; ga.ldo = c.latent_size; ga.round16 = 0; ga.out_f32 = 1;
    ga.wt = use_f16(h) ? h->w1st_s : h->w1st_t;
    if ((rc = launch(h, gc::KC_GEMM_NODE, [&] { return gc::launch_gemm(s, gc::KC_GEMM_NODE, ga, 1, 1, 0, use_f16(h)); }))) return rc;
    ++h->embed_cache_samples;
  }
  auto noisy_write = [&](float sigma_next_call) {
    gc::NoisyWrite nw;
    if (h->fuse_noisy) {
      nw.slots = h->d_slots; nw.c_out = c.c_out; nw.kp = h->kp; nw.scale = f_c_in(std::max(sigma_next_call, 1e-6f));
      if (cache) { nw.xn = h->d_xn; nw.ldn = h->nwp; }
      else nw.xp = h->d_xp;
    }
    return nw;
  };
  auto churned = [&](int i) { return churn && i < n && h->churn_rates[i] > 0.f; };
  // x0 = noise * sigma_0  (dpm_solver_plus_plus_2s.py:71-78)
  bool written = false;                          // the packed input already holds c_in * (state of the next call)
  {
    const gc::NoisyWrite nw = (n > 0 && !churned(0)) ? noisy_write(sigmas[0]) : gc::NoisyWrite();
    if ((rc = launch(h, gc::KC_PACK, [&] { return gc::launch_scale(s, h->d_noise, sigmas[0], ne, h->d_sx, nw); })))
      return rc;
    written = nw.active();
  }
  if (multi) {
    gc::SigmaList sl{};
    for (size_t i = 0; i < call_sigma.size(); ++i) sl.v[i] = call_sigma[i];
    if ((rc = launch(h, gc::KC_COND, [&] {
           return gc::launch_cond_multi(s, sl, (int)call_sigma.size(), c.batch, h->d_nw0t, h->d_nb0, h->d_nw1t,
                                        h->d_nb1, c.noise_num_frequencies, c.noise_hidden, c.noise_base_period,
                                        h->d_wc_all, h->d_bc_all, h->cond_total, h->d_cond_all);
         })))
      return rc;
  }
  auto denoise = [&](const float* x, float sigma) -> int {
    const float ss = std::max(sigma, 1e-6f);  // :84-85
    if (!written) {
      int r = launch(h, gc::KC_PACK, [&] {
        return cache ? gc::launch_write_noisy_compact(s, x, rows, c.c_out, h->nwp, f_c_in(ss), h->d_xn)
                     : gc::launch_write_noisy(s, x, h->d_slots, rows, c.c_out, h->kp, f_c_in(ss), h->d_xp);
      });
      if (r) return r;
    }
    written = false;
    const float* ready = nullptr;
    if (multi) {
      if (calls >= (int)call_sigma.size() || call_sigma[calls] != ss)
        return fail(h, GC_ERR_INTERNAL, "sampler: noise-level list out of step with the loop");
      ready = h->d_cond_all + (size_t)calls * cond_call;
    }
    ++calls;
    return forward(h, ss, ready);
  };
  for (int i = 0; i < n; ++i) {
    float sg = sigmas[i];
    const float sn = sigmas[i + 1];
    if (churn && h->churn_rates[i] > 0.f) {
      // apply_stochastic_churn (gencast/samplers_utils.py:434-452; called at dpm_solver_plus_plus_2s.py:128-137):
      // x <- x + spherical white noise * sqrt(max(s'^2 - s^2, 0)) * inflation, s' = s (1 + rate); the step
      // then runs from s'
      const float s_new = sg * (1.0f + h->churn_rates[i]);
      const float extra = std::sqrt(std::max(s_new * s_new - sg * sg, 0.0f)) * h->churn_inflation;
      if ((rc = noise_field(h, h->d_sx, extra, h->d_sx))) return rc;
      sg = s_new;
    }
    const float sm = std::sqrt(sg * sn);
    if ((rc = denoise(h->d_sx, sg))) return rc;
    const float ss = std::max(sg, 1e-6f);
    const float a_mid = sm / sg;
    {
      const bool mid_call = (sn != 0.0f) || !skip_dead;              // the mid-point state is denoised next
      const gc::NoisyWrite nw = mid_call ? noisy_write(sm) : gc::NoisyWrite();
      if ((rc = launch(h, gc::KC_PACK, [&] {
             return gc::launch_dpm_first(s, h->d_y, h->d_sx, f_c_out(ss), f_c_skip(ss), a_mid, ne,
                                         h->d_sden, h->d_smid, nw);
           })))
        return rc;
      written = nw.active();
    }
    if (sn == 0.0f) {
      // where(sigma_next == 0, x_denoised, x_next) (:148-153): the mid-point call is dead.
      if (!skip_dead && (rc = denoise(h->d_smid, sm))) return rc;
      GC_HIP(h, hipMemcpyAsync(h->d_sx, h->d_sden, ne * sizeof(float), hipMemcpyDeviceToDevice, s));
      written = false;
      continue;
    }
    if ((rc = denoise(h->d_smid, sm))) return rc;
    const float sms = std::max(sm, 1e-6f);
    const float a_next = sn / sg;
    {
      const gc::NoisyWrite nw = (i + 1 < n && !churned(i + 1)) ? noisy_write(sigmas[i + 1]) : gc::NoisyWrite();
      if ((rc = launch(h, gc::KC_PACK, [&] {
             return gc::launch_dpm_second(s, h->d_y, h->d_smid, f_c_out(sms), f_c_skip(sms), a_next, ne,
                                          h->d_sx, nw);
           })))
        return rc;
      written = nw.active();
    }
  }
  *calls_out = calls;
  return GC_OK;
}

void destroy_sample_graph(gc_handle::SampleGraph& g);

int run_sampler(gc_handle* h, const float* sigmas, int n, int skip_dead, gc_sample_stats* stats) {
  const gc_config& c = h->cfg;
  const size_t ne = (size_t)h->hg.G * c.batch * c.c_out;
  hipStream_t s = h->stream;
  int rc, calls = 0;
  const unsigned long long stream0 = h->nz_stream;   // churn noise of this sample starts here
  const bool churn = !h->churn_rates.empty();
  if (churn && (int)h->churn_rates.size() != n)
    return fail(h, GC_ERR_INVALID_ARGUMENT, "gc_set_churn was given a schedule of another length than this sample");
  // The noise level of every denoiser call is known before the loop (churn included): all conditioning
  // vectors of the sample come from ONE launch instead of one per call.
  std::vector<float> call_sigma;
  for (int i = 0; i < n; ++i) {
    float sg = sigmas[i];
    if (churn && h->churn_rates[i] > 0.f) sg = sg * (1.0f + h->churn_rates[i]);
    call_sigma.push_back(std::max(sg, 1e-6f));
    const float sn = sigmas[i + 1];
    if (sn != 0.0f || !skip_dead) call_sigma.push_back(std::max(std::sqrt(sg * sn), 1e-6f));
  }
  const bool multi = (int)call_sigma.size() <= gc::kMaxSigmaList;
  const size_t cond_call = (size_t)c.batch * h->cond_total;
  if (multi && call_sigma.size() * cond_call > h->cond_all_cap) {
    if ((rc = dev_alloc(h, &h->d_cond_all, call_sigma.size() * cond_call))) return rc;
    h->cond_all_cap = call_sigma.size() * cond_call;
  }

  // ---- HIP-graph replay (the reference runs its whole sampler as ONE compiled program: the jax.lax.fori_loop of
  // dpm_solver_plus_plus_2s.py:157-158).  One sample is ~3 500 kernel launches that cost the host ~25 ms to enqueue
  // for ~50 ms of GPU work (nano), which caps how many members one thread can keep in flight.  A sample's launch
  // sequence depends only on its signature -- the noise levels (baked into kernel arguments), skip_dead, the
  // precision / feature mode and which of the two noise buffers it starts from -- so the SECOND sample with a
  // signature is captured (hipStreamBeginCapture on the handle's stream; the first ran eagerly and did every lazy
  // one-time set-up) and every later one is a single hipGraphLaunch.  Eager always: stochastic churn (its noise
  // stream counter is a kernel argument that changes per sample), per-class profiling (events between launches),
  // the debug stops.  Same kernels, same arguments, same order: samples are bit-identical to the eager path.
  const bool eligible = h->use_graphs && multi && !churn && h->prof_cls < 0 && h->debug_layer_limit < 0 &&
                        h->debug_stop_layer < 0 && !h->side_stream;
  gc_handle::SampleGraph* sg = nullptr;
  if (eligible) {
    const bool f16 = use_f16(h), st16 = store16_ok(h);
    for (auto& g : h->sample_graphs)
      if (g.skip_dead == skip_dead && g.noise == h->d_noise && g.f16 == f16 && g.feat16 == h->feat16 && g.st16 == st16 &&
          (int)g.sigmas.size() == n + 1 && !std::memcmp(g.sigmas.data(), sigmas, (n + 1) * sizeof(float)))
        sg = &g;
    if (!sg) {
      if (h->sample_graphs.size() >= 8) {            // forget the least recently used signature
        size_t lru = 0;
        for (size_t i = 1; i < h->sample_graphs.size(); ++i)
          if (h->sample_graphs[i].last_use < h->sample_graphs[lru].last_use) lru = i;
        destroy_sample_graph(h->sample_graphs[lru]);
        h->sample_graphs.erase(h->sample_graphs.begin() + lru);
      }
      gc_handle::SampleGraph g;
      g.sigmas.assign(sigmas, sigmas + n + 1);
      g.skip_dead = skip_dead; g.noise = h->d_noise; g.f16 = f16; g.feat16 = h->feat16; g.st16 = st16;
      g.last_use = ++h->graph_clock;
      h->sample_graphs.push_back(g);
      sg = nullptr;                                  // first sight: eager
    } else {
      sg->last_use = ++h->graph_clock;
    }
  }
  GC_HIP(h, hipEventRecord(h->ev0, s));
  // Threads and graphs (round 4; the record is in DESIGN.md section 5, "the six-thread hang").  What hung once in
  // round 3 was SIX host threads, each inside its handle's first graph call at the same moment: stream capture of
  // ~3 500 launches + hipGraphInstantiate, concurrently, on ROCm 7.2.  Replays alone never did (two threads replaying
  // captured graphs, and one thread driving six handles, ran clean; a hipGraphLaunch behind a running instance returns
  // in 0.06 ms: profiles/r03_graph_probe_nano20.txt).  So exactly that is serialised: capture + instantiate hold a
  // process-wide mutex; launches hold nothing.  Independently of it no executable is launched while its previous
  // instance may still run (two executables per signature, alternated, each behind its own event), and nothing lazy
  // is left for a capturing thread to do: the first, eager sample of a signature has made every one-time runtime
  // call (dynamic-LDS attributes are per device, not per thread; allocations; static switches).
  static std::mutex capture_mutex;
  // Field fallback: the cause of the round-3 hang was found by elimination, not observed in a debugger.
  // GC_TUNE_GRAPH_SERIALIZE=1 restores round 3's wider serialisation -- every hipGraphLaunch also takes the mutex --
  // without a rebuild, should concurrent launch + capture ever misbehave on another ROCm.
  static const bool serialize_launches = [] { const char* v = std::getenv("GC_TUNE_GRAPH_SERIALIZE"); return v && *v == '1'; }();
  auto launch_exec = [&](gc_handle::SampleGraph* g) -> int {
    std::unique_lock<std::mutex> launch_lock(capture_mutex, std::defer_lock);
    if (serialize_launches) launch_lock.lock();
    const int i = g->exec2 ? g->next : 0;
    hipGraphExec_t ex = i ? g->exec2 : g->exec;
    if (!g->done[i]) GC_HIP(h, hipEventCreateWithFlags(&g->done[i], hipEventDisableTiming));
    else GC_HIP(h, hipEventSynchronize(g->done[i]));   // its previous instance has ended (usually long ago)
    GC_HIP(h, hipGraphLaunch(ex, s));
    GC_HIP(h, hipEventRecord(g->done[i], s));
    g->next = i ^ 1;
    return GC_OK;
  };
  if (sg && sg->exec) {
    if ((rc = launch_exec(sg))) return rc;
    calls = sg->calls;
    h->launches_last_call = sg->launches_per_call;
    h->st16 = h->last_st16 = sg->st16;
    ++h->graph_replays;
  } else if (sg) {
    const int64_t l0 = h->launch_count;
    static const bool verbose = [] { const char* v = std::getenv("GC_TUNE_GRAPH_VERBOSE"); return v && *v == '1'; }();
    auto say = [&](const char* what) { if (verbose) { std::fprintf(stderr, "[gc graph] %s\n", what); std::fflush(stderr); } };
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr, exec2 = nullptr;
    {
      std::lock_guard<std::mutex> capture_lock(capture_mutex);
      say("begin capture");
      GC_HIP(h, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      rc = sampler_body(h, sigmas, n, skip_dead, call_sigma, multi, &calls);
      const hipError_t e_end = hipStreamEndCapture(s, &graph);
      say("end capture");
      if (rc) {
        if (graph) (void)hipGraphDestroy(graph);
        return rc;
      }
      if (e_end != hipSuccess || !graph) return fail(h, GC_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e_end));
      hipError_t e_inst = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
      if (e_inst == hipSuccess) e_inst = hipGraphInstantiate(&exec2, graph, nullptr, nullptr, 0);
      if (e_inst != hipSuccess) {
        if (exec) (void)hipGraphExecDestroy(exec);
        (void)hipGraphDestroy(graph);
        return fail(h, GC_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e_inst));
      }
      say("instantiated");
    }
    sg->graph = graph; sg->exec = exec; sg->exec2 = exec2; sg->calls = calls; sg->next = 0;
    sg->launches = h->launch_count - l0;
    sg->launches_per_call = h->launches_last_call;
    ++h->graph_captures;
    if ((rc = launch_exec(sg))) return rc;
    say("launched");
    ++h->graph_replays;
  } else if ((rc = sampler_body(h, sigmas, n, skip_dead, call_sigma, multi, &calls))) {
    return rc;
  }
  GC_HIP(h, hipEventRecord(h->ev1, s));
  h->has_sample = true;
  // domain guard: NaN / Inf stick to a sample row once they appear, so one check of the final sample
  // covers all 39 calls; it is resolved at the next synchronising entry point (resolve_guard)
  if (use_f16(h)) {
    if ((rc = guard_enqueue(h, h->d_sx, ne))) return rc;
    h->last_sigmas.assign(sigmas, sigmas + n + 1);
    h->last_skip_dead = skip_dead;
    h->guard_pending = true;
    h->last_stream0 = stream0;
    h->last_noise = h->d_noise;
  }
  if (stats) {
    GC_HIP(h, hipEventSynchronize(h->ev1));
    float ms = 0.f;
    GC_HIP(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    stats->denoiser_calls = calls;
    stats->device_ms = ms;
  }
  return GC_OK;
}

// Waits for the stream; if the last resident sample left the f16x3 domain (its output holds NaN /
// Inf), samples again with the exact-f32 kernels from the same noise and conditioning.
int resolve_guard(gc_handle* h) {
  GC_HIP(h, hipStreamSynchronize(h->stream));
  if (!h->guard_pending) return GC_OK;
  h->guard_pending = false;
  if (!guard_tripped(h)) return GC_OK;
  ++h->range_fallbacks;
  h->in_fallback = true;
  const unsigned long long stream_end = h->nz_stream;
  h->nz_stream = h->last_stream0;                    // the re-run draws the same churn noise
  const std::vector<float> sig = h->last_sigmas;
  float* const noise_next = h->d_noise;              // the re-run starts from the noise of the sample it repeats
  h->d_noise = h->last_noise;
  int rc = run_sampler(h, sig.data(), (int)sig.size() - 1, h->last_skip_dead, nullptr);
  h->d_noise = noise_next;
  h->nz_stream = stream_end;
  h->in_fallback = false;
  if (rc) return rc;
  GC_HIP(h, hipStreamSynchronize(h->stream));
  return GC_OK;
}

void destroy_sample_graph(gc_handle::SampleGraph& g) {
  for (hipEvent_t& e : g.done)                       // an executable is destroyed only after its last launch has ended
    if (e) { (void)hipEventSynchronize(e); (void)hipEventDestroy(e); e = nullptr; }
  if (g.exec) (void)hipGraphExecDestroy(g.exec);
  if (g.exec2) (void)hipGraphExecDestroy(g.exec2);
  if (g.graph) (void)hipGraphDestroy(g.graph);
  g.exec = g.exec2 = nullptr;
  g.graph = nullptr;
}

// Captured sampler graphs bake device pointers and launch geometry: dropped whenever those may change.
void drop_sample_graphs(gc_handle* h) {
  for (auto& g : h->sample_graphs) destroy_sample_graph(g);
  h->sample_graphs.clear();
}

// Entry points that overwrite the initial noise while a sample's domain check is pending write the OTHER buffer.
void protect_pending_noise(gc_handle* h) {
  if (h->guard_pending && h->d_noise == h->last_noise) std::swap(h->d_noise, h->d_noise_alt);
}

// Asynchronous H2D through a handle-owned pinned buffer: the caller's buffer is free on return.
int staged_upload(gc_handle* h, float* pinned, float* dev, const float* src, size_t count) {
  GC_HIP(h, hipEventSynchronize(h->ev_pin));       // the previous copy out of a staging buffer is done
  std::memcpy(pinned, src, count * sizeof(float));
  GC_HIP(h, hipMemcpyAsync(dev, pinned, count * sizeof(float), hipMemcpyHostToDevice, h->stream));
  GC_HIP(h, hipEventRecord(h->ev_pin, h->stream));
  return GC_OK;
}

int check_ready(gc_handle* h) {
  if (!h) return GC_ERR_INVALID_ARGUMENT;
  if (!h->finalized) return fail(h, GC_ERR_STATE, "gc_finalize has not been called");
  return GC_OK;
}


// ---- RCCL, bound at run time ----------------------------------------------------------------------
// The denoiser itself never communicates; only the ensemble driver's one exchange per forecast step
// does.  librccl is therefore not a link-time dependency: a single-GPU user (or a CPU-only box that
// only checks the ABI) never loads it.
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};

Rccl& rccl() {
  static Rccl r = [] {
    Rccl x;
    const char* override_path = std::getenv("GC_RCCL_LIBRARY");
    // ROCm's own library first: by search path the name resolved to the copy inside torch's wheel on a test box
    // (profiles/r03_gpu_tests.log); GC_RCCL_LIBRARY overrides
    for (const char* name : {override_path ? override_path : "/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"}) {
      x.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (x.lib) break;
    }
    if (!x.lib) {
      x.error = std::string("cannot load librccl: ") + dlerror();
      return x;
    }
    auto sym = [&](const char* n) {
      void* p = dlsym(x.lib, n);
      if (!p && x.error.empty()) x.error = std::string("librccl lacks ") + n;
      return p;
    };
    x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(sym("ncclGetUniqueId"));
    x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(sym("ncclCommInitRank"));
    x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(sym("ncclCommDestroy"));
    x.CommCount = reinterpret_cast<decltype(x.CommCount)>(sym("ncclCommCount"));
    x.CommUserRank = reinterpret_cast<decltype(x.CommUserRank)>(sym("ncclCommUserRank"));
    x.Broadcast = reinterpret_cast<decltype(x.Broadcast)>(sym("ncclBroadcast"));
    x.AllReduce = reinterpret_cast<decltype(x.AllReduce)>(sym("ncclAllReduce"));
    x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(sym("ncclGetErrorString"));
    return x;
  }();
  return r;
}

#define GC_NCCL(h, call)                                                                    \
  do {                                                                                      \
    ncclResult_t r__ = (call);                                                              \
    if (r__ != ncclSuccess) {                                                               \
      (h)->err = std::string(#call) + ": " + rccl().GetErrorString(r__);                    \
      return GC_ERR_COMM;                                                                   \
    }                                                                                       \
  } while (0)

// No C++ exception crosses the C ABI: every entry point runs inside this wrapper.
template <typename F>
int guarded(gc_handle* h, F&& f) noexcept {
  const char* what = "unknown C++ exception";
  try {
    return f();
  } catch (const std::bad_alloc&) {
    what = "out of host memory (std::bad_alloc)";
  } catch (const std::exception& e) {
    try {
      (h ? h->err : g_create_error) = std::string("C++ exception: ") + e.what();
      return GC_ERR_INTERNAL;
    } catch (...) {
    }
  } catch (...) {
  }
  try {
    (h ? h->err : g_create_error) = what;
  } catch (...) {
  }
  return GC_ERR_INTERNAL;
}

}  // namespace

// =================================================================================================
static void destroy_impl(gc_handle* h);

extern "C" {

int gc_abi_version(void) { return GC_ABI_VERSION; }

const char* gc_build_info(void) {
#ifndef GC_SOURCE_HASH
#define GC_SOURCE_HASH "unknown"
#endif
  return "libgencast_hip gfx950 f16x3/f32-mfma " __DATE__ " " __TIME__ " src:" GC_SOURCE_HASH;
}

int gc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int gc_device_pci_bus_id(int32_t device_id, char* out, int64_t cap) {
  if (!out || cap < 16 || device_id < 0 || device_id >= gc_device_count()) return GC_ERR_INVALID_ARGUMENT;
  out[0] = 0;
  return hipDeviceGetPCIBusId(out, (int)cap, device_id) == hipSuccess ? GC_OK : GC_ERR_HIP;
}

const char* gc_last_error(const gc_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int gc_create(const gc_config* cfg, int device_id, gc_handle** out) {
  return guarded(nullptr, [&]() -> int {
  if (!cfg || !out) { g_create_error = "null argument"; return GC_ERR_INVALID_ARGUMENT; }
  *out = nullptr;
  const gc_config& c = *cfg;
  auto bad = [&](const char* m) { g_create_error = m; return GC_ERR_INVALID_ARGUMENT; };
  if (c.latent_size <= 0 || c.d_model <= 0 || c.num_heads <= 0 || c.ffw_hidden <= 0 ||
      c.num_layers < 0 || c.c_in <= 0 || c.c_out <= 0 || c.batch <= 0)
    return bad("all dimensions must be positive");
  if (c.c_out > c.c_in) return bad("c_out cannot exceed c_in (noisy targets are part of the forcings)");
  if (c.latent_size != c.d_model) return bad("latent_size must equal d_model");
  if (c.d_model % c.num_heads) return bad("num_heads has to divide d_model exactly");
  if (c.noise_num_frequencies <= 0 || c.noise_num_frequencies > 128 || c.noise_hidden <= 0 ||
      c.noise_hidden > 128 || !(c.noise_base_period > 0))
    return bad("noise encoder sizes out of range");
  auto unsup = [&](const char* m) { g_create_error = m; return GC_ERR_UNSUPPORTED; };
  if (c.latent_size != 128 && c.latent_size != 256 && c.latent_size != 512)
    return unsup("latent_size must be 128, 256 or 512 (MFMA column tiling)");
  const int dh = c.d_model / c.num_heads;
  if (dh != 32 && dh != 64 && dh != 128) return unsup("head size must be 32, 64 or 128");
  if (c.num_heads > 8) return unsup("at most 8 heads");
  if (c.ffw_hidden % 128) return unsup("ffw_hidden must be a multiple of 128");
  if (c.c_out > 512) return unsup("c_out must be <= 512");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    g_create_error = "no HIP device visible (this library has no CPU fallback)";
    return GC_ERR_NO_DEVICE;
  }
  if (device_id < 0 || device_id >= ndev) { g_create_error = "device_id out of range"; return GC_ERR_NO_DEVICE; }
  std::unique_ptr<gc_handle> h(new gc_handle());
  h->cfg = c;
  h->device = device_id;
  hipError_t e = hipSetDevice(device_id);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreate(&h->ev0);
  if (e == hipSuccess) e = hipEventCreate(&h->ev1);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_pin, hipEventDisableTiming);
  if (e == hipSuccess) e = hipMalloc((void**)&h->d_nonfinite, sizeof(unsigned));
  if (e == hipSuccess) e = hipMemset(h->d_nonfinite, 0, sizeof(unsigned));
  if (e == hipSuccess) e = hipHostMalloc((void**)&h->h_nonfinite, sizeof(unsigned), hipHostMallocDefault);
  if (e != hipSuccess) {
    g_create_error = hipGetErrorString(e);
    gc_destroy(h.release());
    return GC_ERR_HIP;
  }
  *h->h_nonfinite = 0;
  h->kp = round_up(3 + c.c_in, 32);
  {
    const char* pv = std::getenv("GC_PRECISION");
    h->f16x3 = !(pv && std::string(pv) == "f32");
    const char* gv = std::getenv("GC_TUNE_GRAPH");
    h->use_graphs = !(gv && std::string(gv) == "0");
  }
  build_specs(h.get());
  *out = h.release();
  return GC_OK;
  });
}

void gc_destroy(gc_handle* h) {
  if (!h) return;
  try {
    destroy_impl(h);
  } catch (...) {
  }
}

static void destroy_impl(gc_handle* h) {
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  drop_sample_graphs(h);
  if (h->comm) (void)rccl().CommDestroy(h->comm);
  for (void* p : h->allocs) (void)hipFree(p);
  if (h->d_nonfinite) (void)hipFree(h->d_nonfinite);
  for (void* p : {(void*)h->h_nonfinite, (void*)h->pin_cond, (void*)h->pin_noise, (void*)h->pin_forc})
    if (p) (void)hipHostFree(p);
  if (h->ev_pin) (void)hipEventDestroy(h->ev_pin);
  if (h->ev_stash) (void)hipEventDestroy(h->ev_stash);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->stream2) {
    (void)hipStreamSynchronize(h->stream2);
    (void)hipStreamDestroy(h->stream2);
  }
  for (hipEvent_t e : h->prof_events) (void)hipEventDestroy(e);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int gc_set_option(gc_handle* h, const char* key, const char* value) {
  return guarded(h, [&]() -> int {
  if (!h) return GC_ERR_INVALID_ARGUMENT;
  if (!key || !value) return fail(h, GC_ERR_INVALID_ARGUMENT, "null argument");
  const std::string k(key), v(value);
  // A pending exact-f32 re-run of the LAST sample (f16x3 domain guard) belongs to the mode that sample was drawn in:
  // it is resolved BEFORE a flag changes (the flags are part of the forward's identity -- fp16 storage, graph signature).
  auto settle = [&]() -> int {
    if (!h->finalized) return GC_OK;
    GC_HIP(h, hipSetDevice(h->device));
    return resolve_guard(h);
  };
  if (k == "precision") {
    if (v != "f16x3" && v != "f32") return fail(h, GC_ERR_INVALID_ARGUMENT, "precision must be f16x3 or f32");
    const bool want = v == "f16x3";
    if (want == h->f16x3) return GC_OK;
    if (int rc = settle()) return rc;
    h->f16x3 = want;
    return h->finalized ? compute_static_embeddings(h) : GC_OK;   // the static embeddings follow the precision
  }
  if (k == "hidden_layers") {
    // DenoiserArchitectureConfig.hidden_layers (gencast/denoiser.py:108,135,374,402 -> common/mlp.py:157-199): hidden
    // layers of every MLP of the two GNNs.  It fixes the parameter names, so it is set before any weight is loaded.
    char* end = nullptr;
    const long n = std::strtol(v.c_str(), &end, 10);
    if (v.empty() || (end && *end) || n < 1 || n > 4) return fail(h, GC_ERR_INVALID_ARGUMENT, "hidden_layers must be in 1..4");
    if ((int)n == h->hidden_layers) return GC_OK;
    if (h->finalized || !h->weights.empty())
      return fail(h, GC_ERR_STATE, "hidden_layers must be set before the first gc_load_weight");
    h->hidden_layers = (int)n;
    h->specs.clear();
    build_specs(h);
    return GC_OK;
  }
  if (k == "features") {
    if (v != "f16" && v != "f32") return fail(h, GC_ERR_INVALID_ARGUMENT, "features must be f32 or f16");
    const bool want = v == "f16";
    if (want == h->feat16) return GC_OK;
    if (int rc = settle()) return rc;
    h->feat16 = want;
    return h->finalized ? compute_static_embeddings(h) : GC_OK;   // their internal roundings follow the mode
  }
  if (k == "grid2mesh_aggregate_normalization") {
    // DenoiserArchitectureConfig.grid2mesh_aggregate_normalization (gencast/denoiser.py:123,138,367): the summed
    // grid2mesh edge messages of every mesh node are divided by this constant (deep_typed_graph_net.py:396-410);
    // "0" or "none": not normalised (the reference default)
    char* end = nullptr;
    const float f = (v == "none" || v.empty()) ? 0.f : std::strtof(v.c_str(), &end);
    if ((end && *end) || !(f >= 0.f) || !std::isfinite(f))
      return fail(h, GC_ERR_INVALID_ARGUMENT, "grid2mesh_aggregate_normalization must be a non-negative number");
    if (f == h->g2m_agg_norm) return GC_OK;
    if (int rc = settle()) return rc;
    h->g2m_agg_norm = f;
    drop_sample_graphs(h);                       // the constant is a kernel argument baked into captured samples
    return GC_OK;
  }
  if (k == "graphs") {
    if (v == "on") h->use_graphs = true;
    else if (v == "off") h->use_graphs = false;
    else return fail(h, GC_ERR_INVALID_ARGUMENT, "graphs must be on or off");
    return GC_OK;
  }
  return fail(h, GC_ERR_INVALID_ARGUMENT, "unknown option: " + k);
  });
}

int gc_set_graph(gc_handle* h, int32_t G, int32_t M, int32_t E1, const int32_t* g2m_s,
                 const int32_t* g2m_r, int32_t E2, const int32_t* m2g_s, const int32_t* m2g_r,
                 const int32_t* khop_rowptr, const int32_t* khop_cols, const float* grid_struct,
                 const float* mesh_struct, const float* g2m_edge_struct, const float* m2g_edge_struct,
                 const float* mesh_xyz) {
  return guarded(h, [&]() -> int {
  if (!h) return GC_ERR_INVALID_ARGUMENT;
  if (h->has_graph) return fail(h, GC_ERR_STATE, "graph already set on this handle");
  if (!g2m_s || !g2m_r || !m2g_s || !m2g_r || !khop_rowptr || !khop_cols || !grid_struct ||
      !mesh_struct || !g2m_edge_struct || !m2g_edge_struct)
    return fail(h, GC_ERR_INVALID_ARGUMENT, "null graph array");
  std::string msg = gc::build_host_graph(G, M, E1, g2m_s, g2m_r, E2, m2g_s, m2g_r, khop_rowptr,
                                         khop_cols, mesh_xyz, &h->hg);
  if (!msg.empty()) return fail(h, GC_ERR_INVALID_ARGUMENT, msg);
  GC_HIP(h, hipSetDevice(h->device));
  const gc::HostGraph& g = h->hg;
  int rc;
  if ((rc = dev_upload(h, &h->d_g2m_snd, g.g2m_snd))) return rc;
  if ((rc = dev_upload(h, &h->d_g2m_rcv, g.g2m_rcv))) return rc;
  if ((rc = dev_upload(h, &h->d_m2g_snd, g.m2g_snd))) return rc;
  if ((rc = dev_upload(h, &h->d_m2g_rcv, g.m2g_rcv))) return rc;
  if ((rc = dev_upload(h, &h->d_g2m_ptr, g.g2m_ptr))) return rc;
  if ((rc = dev_upload(h, &h->d_g2m_eid, g.g2m_eid))) return rc;
  if ((rc = dev_upload(h, &h->d_m2g_ptr, g.m2g_ptr))) return rc;
  if ((rc = dev_upload(h, &h->d_m2g_eid, g.m2g_eid))) return rc;
  if ((rc = dev_upload(h, &h->d_tile_start, g.tile_chunk_start))) return rc;
  {
    std::vector<int> items, tiles;
    const char* sw = std::getenv("GC_TUNE_ATTN_ITEMS");
    h->att_n_items = 0;
    if (!(sw && *sw && std::atoi(sw) == 0) && h->cfg.batch == 1 && h->cfg.num_heads <= 4 &&
        h->cfg.d_model / h->cfg.num_heads == 128 && build_attention_items(g, &items, &tiles)) {
      if ((rc = dev_upload(h, &h->d_att_items, items))) return rc;
      if ((rc = dev_upload(h, &h->d_att_tiles, tiles))) return rc;
      h->att_n_items = (int)(items.size() / 4);
    }
  }
  h->max_tile_chunks = 0;
  for (int t = 0; t < g.n_tiles; ++t)
    h->max_tile_chunks = std::max(h->max_tile_chunks, g.tile_chunk_start[t + 1] - g.tile_chunk_start[t]);
  if ((rc = dev_upload(h, &h->d_union, g.union_idx))) return rc;
  if ((rc = dev_upload(h, &h->d_mask, g.mask_bits))) return rc;
  if ((rc = dev_upload(h, &h->d_grid_struct, std::vector<float>(grid_struct, grid_struct + (size_t)G * 3)))) return rc;
  std::vector<float> ms16((size_t)M * 32, 0.f), e1s((size_t)E1 * 32, 0.f), e2s((size_t)E2 * 32, 0.f);
  for (int i = 0; i < M; ++i)           // internal mesh order
    for (int k = 0; k < 3; ++k) ms16[(size_t)i * 32 + k] = mesh_struct[(size_t)g.perm[i] * 3 + k];
  for (int e = 0; e < E1; ++e)
    for (int k = 0; k < 4; ++k) e1s[(size_t)e * 32 + k] = g2m_edge_struct[(size_t)e * 4 + k];
  for (int e = 0; e < E2; ++e)
    for (int k = 0; k < 4; ++k) e2s[(size_t)e * 32 + k] = m2g_edge_struct[(size_t)g.m2g_order[e] * 4 + k];   // internal edge order
  if ((rc = dev_upload(h, &h->d_mesh_struct16, ms16))) return rc;
  if ((rc = dev_upload(h, &h->d_e1_struct16, e1s))) return rc;
  if ((rc = dev_upload(h, &h->d_e2_struct16, e2s))) return rc;

  // activations
  const gc_config& c = h->cfg;
  const size_t B = c.batch, L = c.latent_size, D = c.d_model, F = c.ffw_hidden;
  const size_t GB = (size_t)G * B, MB = (size_t)M * B;
  if ((rc = dev_alloc(h, &h->d_sigma, B))) return rc;
  if ((rc = dev_alloc(h, &h->d_condvec, B * gc::kCondDim))) return rc;
  if ((rc = dev_alloc(h, &h->d_feats, GB * c.c_in))) return rc;
  if ((rc = dev_alloc(h, &h->d_xp, GB * h->kp))) return rc;
  if ((rc = dev_alloc(h, &h->d_g0, GB * L))) return rc;
  if ((rc = dev_alloc(h, &h->d_g1, GB * L))) return rc;
  if ((rc = dev_alloc(h, &h->d_g2, GB * L))) return rc;
  if ((rc = dev_alloc(h, &h->d_agg2, GB * L))) return rc;
  if ((rc = dev_alloc(h, &h->d_m0, MB * L))) return rc;
  if ((rc = dev_alloc(h, &h->d_x, MB * L))) return rc;
  if ((rc = dev_alloc(h, &h->d_agg1, MB * L))) return rc;
  if ((rc = dev_alloc(h, &h->d_m2, MB * L))) return rc;
  if ((rc = dev_alloc(h, &h->d_qkv, MB * 3 * D))) return rc;
  if ((rc = dev_alloc(h, &h->d_att, MB * D))) return rc;
  {
    uint16_t* kv = nullptr;
    if ((rc = dev_alloc(h, &kv, MB * 4 * D))) return rc;
    h->d_kv16 = kv;
  }
  if ((rc = dev_alloc(h, &h->d_u, MB * F))) return rc;
  {
    auto env_int = [](const char* name, int dflt) {
      const char* v = std::getenv(name);
      return (v && *v) ? std::atoi(v) : dflt;
    };
    auto largest_split = [](int k, int cap) {
      int best = 1;
      for (int sp = 1; sp <= cap; ++sp)
        if (k % (32 * sp) == 0) best = sp;
      return best;
    };
    // enough attention blocks to cover the 256 CUs about once
    int as = (int)std::lround(256.0 / std::max(1, h->hg.n_tiles * (int)B));
    h->attn_splits = std::min(8, std::max(1, env_int("GC_TUNE_ATTN_SPLITS", as)));
    // FFW layer 2 as a GEMM of its own (d_model 512, or the fused FFW switched off): K = ffw_hidden is split only while
    // the unsplit launch would leave CUs idle -- every split is one more float32 slab the next row pass reads.  At the
    // 1-degree size (161 row tiles x 4 column panels = 644 workgroups) 1 split vs 4: FFW-2 1.39 -> 1.33 ms and the row
    // pass 0.44 -> 0.30 ms per call, +3.2 % calls/s (float32 features), +2.5 % (fp16 features).
    const int ffw2_wgs = (int)((MB + 63) / 64) * (int)(D / 128);
    h->ffw2_splits = largest_split((int)F, std::max(1, env_int("GC_TUNE_FFW2_SPLITS", ffw2_wgs >= 512 ? 1 : 4)));
    h->out_splits = largest_split((int)D, std::max(1, env_int("GC_TUNE_OUT_SPLITS", 2)));
    h->mt_qkv = env_int("GC_TUNE_MT_QKV", 1) == 2 ? 2 : 1;
    h->mt_out = env_int("GC_TUNE_MT_OUT", 1) == 2 ? 2 : 1;
    h->mt_ffw1 = env_int("GC_TUNE_MT_FFW1", 1) == 2 ? 2 : 1;
    h->mt_ffw2 = env_int("GC_TUNE_MT_FFW2", 1) == 2 ? 2 : 1;
    // both FFW layers in one launch (f16x3 weight-streaming form; GC_TUNE_FFW_FUSED=0 for the two-launch form)
    // (at d_model = 512 the fused kernel's accumulators leave one workgroup per CU and the two-launch
    //  form is faster: 2.98 vs 3.32 ms per call on the 1-degree config; GC_TUNE_FFW_FUSED=2 forces it)
    h->fuse_combine = env_int("GC_TUNE_FUSE_COMBINE", 1) != 0;
    h->gemm_ws = env_int("GC_TUNE_GEMM_WS", 1) != 0;
    h->ffw_xcd = env_int("GC_TUNE_FFW_XCD", 0);
    h->f32_ws = env_int("GC_TUNE_F32_WS", 1) != 0 && h->gemm_ws && D % 128 == 0 && F % 256 == 0;
    h->fuse_outrow = env_int("GC_TUNE_FUSE_OUTROW", 1) != 0;
    h->attn_f16 = env_int("GC_TUNE_ATTN_F16", 1) != 0;
    h->attn_v2 = env_int("GC_TUNE_ATTN_V2", 1) != 0;
    h->fuse_noisy = env_int("GC_TUNE_FUSE_NOISY", 1) != 0;
    h->side_stream = env_int("GC_TUNE_SIDE_STREAM", 0) != 0;
    h->wt_stores = env_int("GC_TUNE_WT_STORES", 0);
    h->a16 = env_int("GC_TUNE_A16", 1) != 0;
    h->attn_v2_force = env_int("GC_TUNE_ATTN_V2", 1) == 2;
    h->ws_mt = env_int("GC_TUNE_WS_MT", 0);
    h->mlp_ws = env_int("GC_TUNE_MLP_WS", 1) != 0;
    h->mlp_ws512 = env_int("GC_TUNE_MLP_WS512", 2) != 0;
    h->m2g_fuse_sum = env_int("GC_TUNE_M2G_FUSE_SUM", 1) != 0;
    h->embed_cache = env_int("GC_TUNE_EMBED_CACHE", 1) != 0;
    h->mlp_pair = env_int("GC_TUNE_MLP_PAIR", 1) != 0;
    const int want_fused = env_int("GC_TUNE_FFW_FUSED", 1);
    h->ffw_fused_slabs = (h->gemm_ws && want_fused != 0 && D % 128 == 0 && (D <= 256 || (want_fused == 2 && D <= 512)) &&
                          F % 256 == 0 && F / 256 <= 16) ? (int)(F / 256) : 0;
    const size_t slabs = (size_t)std::max(std::max(h->ffw2_splits, h->out_splits), std::max(h->ffw_fused_slabs, 1));
    if ((rc = dev_alloc(h, &h->d_h, MB * D))) return rc;
    if ((rc = dev_alloc(h, &h->d_pg, GB * L))) return rc;
    if ((rc = dev_alloc(h, &h->d_pm, MB * L))) return rc;
    if (!h->d_ones) {
      if ((rc = dev_upload(h, &h->d_ones, std::vector<float>(2048, 1.0f)))) return rc;
      if ((rc = dev_upload(h, &h->d_zeros, std::vector<float>(2048, 0.0f)))) return rc;
    }
    // Edge MLPs with the first layer split by input block (e @ Wa + (n_s @ Wb)[senders] + (n_r @ Wc)[receivers]: the two
    // node products are computed once per NODE, 46 % of the first layer's FLOPs at the 1-degree sizes).  On the
    // weight-streaming kernels: 1 degree / latent 512 +7.6 % calls/s (float32 features) / +4.3 % (fp16 features);
    // nano / latent 256: -0.7 % (four more launches buy too little there) -- so on from latent 512.
    h->split_edge = env_int("GC_TUNE_SPLIT_EDGE", L >= 512 ? 1 : 0) != 0;
    if ((rc = dev_alloc(h, &h->d_part, slabs * MB * D))) return rc;
    const size_t aslots = (size_t)h->hg.n_tiles * h->attn_splits * B * c.num_heads;
    if ((rc = dev_alloc(h, &h->d_apart_o, aslots * 32 * (D / c.num_heads)))) return rc;
    if ((rc = dev_alloc(h, &h->d_apart_ml, aslots * 32 * 2))) return rc;
  }
  if ((rc = dev_alloc(h, &h->d_e1, (size_t)E1 * B * L))) return rc;
  if ((rc = dev_alloc(h, &h->d_f1, (size_t)E2 * B * L))) return rc;
  if ((rc = dev_alloc(h, &h->d_y, GB * c.c_out))) return rc;
  if ((rc = dev_alloc(h, &h->d_sx, GB * c.c_out))) return rc;
  if ((rc = dev_alloc(h, &h->d_sden, GB * c.c_out))) return rc;
  if ((rc = dev_alloc(h, &h->d_smid, GB * c.c_out))) return rc;
  if ((rc = dev_alloc(h, &h->d_noise, GB * c.c_out))) return rc;
  if ((rc = dev_alloc(h, &h->d_noise_alt, GB * c.c_out))) return rc;
  if ((rc = dev_alloc(h, &h->d_slots, (size_t)c.c_out))) return rc;
  if ((rc = dev_alloc(h, &h->d_m0_hat, (size_t)M * L))) return rc;
  if ((rc = dev_alloc(h, &h->d_e0_hat, (size_t)E1 * L))) return rc;
  if ((rc = dev_alloc(h, &h->d_f0_hat, (size_t)E2 * L))) return rc;
  GC_HIP(h, hipMemset(h->d_xp, 0, GB * h->kp * sizeof(float)));
  GC_HIP(h, hipHostMalloc((void**)&h->pin_cond, GB * c.c_in * sizeof(float), hipHostMallocDefault));
  GC_HIP(h, hipHostMalloc((void**)&h->pin_noise, GB * c.c_out * sizeof(float), hipHostMallocDefault));
  h->has_graph = true;
  h->finalized = false;
  return GC_OK;
  });
}

int gc_load_weight(gc_handle* h, const char* name, const float* data, const int64_t* shape,
                   int32_t ndim) {
  return guarded(h, [&]() -> int {
  if (!h) return GC_ERR_INVALID_ARGUMENT;
  if (!name || !data || !shape) return fail(h, GC_ERR_INVALID_ARGUMENT, "null argument");
  const std::string n(name);
  const std::string dead = std::string(P_M2G) + ".processor_networks.0.graph_network.update_node_fns.mesh_nodes.";
  if (n.compare(0, dead.size(), dead) == 0) return GC_OK;  // never read by the reference either
  auto it = h->specs.find(n);
  if (it == h->specs.end()) return fail(h, GC_ERR_INVALID_ARGUMENT, "unknown parameter name: " + n);
  const auto& want = it->second;
  bool ok = ((int)want.size() == ndim);
  size_t count = 1;
  for (int i = 0; ok && i < ndim; ++i) { ok = (shape[i] == want[i]); count *= (size_t)shape[i]; }
  if (!ok) {
    std::string w = "(";
    for (size_t i = 0; i < want.size(); ++i) w += (i ? "," : "") + std::to_string(want[i]);
    return fail(h, GC_ERR_INVALID_ARGUMENT, "shape mismatch for " + n + ", expected " + w + ")");
  }
  h->weights[n].assign(data, data + count);
  h->finalized = false;
  h->finalized_weights = false;
  h->embed_cache_ready = false;
  return GC_OK;
  });
}

int gc_missing_weights(gc_handle* h, int32_t* count) {
  return guarded(h, [&]() -> int {
  if (!h || !count) return GC_ERR_INVALID_ARGUMENT;
  int miss = 0;
  for (const auto& kv : h->specs)
    if (!h->weights.count(kv.first)) ++miss;
  *count = miss;
  return GC_OK;
  });
}

int gc_finalize(gc_handle* h) {
  return guarded(h, [&]() -> int {
  if (!h) return GC_ERR_INVALID_ARGUMENT;
  if (!h->has_graph) return fail(h, GC_ERR_STATE, "gc_set_graph must be called before gc_finalize");
  for (const auto& kv : h->specs)
    if (!h->weights.count(kv.first)) return fail(h, GC_ERR_STATE, "missing parameter: " + kv.first);
  GC_HIP(h, hipSetDevice(h->device));
  GC_HIP(h, hipStreamSynchronize(h->stream));
  drop_sample_graphs(h);                        // the weight images below are new allocations
  const gc_config& c = h->cfg;
  const int L = c.latent_size, D = c.d_model, F = c.ffw_hidden;
  const std::string g = P_G2M, m = P_M2G, t = P_TR, nz = P_NOISE;
  CondPacker cp;
  int rc;
  const int node_in = 3 + c.c_in;
  const std::string gn = g + ".processor_networks.0.graph_network";
  const std::string gn2 = m + ".processor_networks.0.graph_network";
  if (h->hidden_layers >= 2) {
    // every MLP is a chain of launches (run_mlp): the algebraic edge-MLP split and the side stream assume ONE launch
    h->split_edge = false;
    h->side_stream = false;
    if (!h->d_mlp_tmp[0]) {
      const size_t rows = (size_t)std::max(std::max(h->hg.G, h->hg.M), std::max(h->hg.E1, h->hg.E2)) * (size_t)c.batch;
      for (int i = 0; i < 2; ++i)
        if ((rc = dev_alloc(h, &h->d_mlp_tmp[i], rows * (size_t)L))) return rc;
    }
  }
  // NOTE: device buffers of a previous gc_finalize stay allocated until gc_destroy.
  if ((rc = upload_mlp(h, g + ".embedder_network.embed_node_fns.grid_nodes", node_in, 0, node_in, h->kp, L, L, true, &cp, &h->g2m_embed_grid))) return rc;
  // mesh nodes see [struct(3) | zeros(c_in)] (denoiser.py:661-668): only the first 3 kernel rows matter.
  if ((rc = upload_mlp(h, g + ".embedder_network.embed_node_fns.mesh_nodes", node_in, 0, 3, 32, L, L, true, &cp, &h->g2m_embed_mesh))) return rc;
  if ((rc = upload_mlp(h, g + ".embedder_network.embed_edge_fns.grid2mesh", 4, 0, 4, 32, L, L, true, &cp, &h->g2m_embed_edge))) return rc;
  if ((rc = upload_mlp(h, gn + ".update_edge_fns.grid2mesh.edge_fn", 3 * L, 0, 3 * L, 3 * L, L, L, true, &cp, &h->g2m_edge))) return rc;
  if ((rc = upload_mlp(h, gn + ".update_node_fns.mesh_nodes.node_fn", 2 * L, 0, 2 * L, 2 * L, L, L, true, &cp, &h->g2m_mesh))) return rc;
  if ((rc = upload_mlp(h, gn + ".update_node_fns.grid_nodes.node_fn", L, 0, L, L, L, L, true, &cp, &h->g2m_grid))) return rc;
  if ((rc = upload_mlp(h, m + ".embedder_network.embed_edge_fns.mesh2grid", 4, 0, 4, 32, L, L, true, &cp, &h->m2g_embed_edge))) return rc;
  if ((rc = upload_mlp(h, gn2 + ".update_edge_fns.mesh2grid.edge_fn", 3 * L, 0, 3 * L, 3 * L, L, L, true, &cp, &h->m2g_edge))) return rc;
  if ((rc = upload_mlp(h, gn2 + ".update_node_fns.grid_nodes.node_fn", 2 * L, 0, 2 * L, 2 * L, L, L, true, &cp, &h->m2g_grid))) return rc;
  if ((rc = upload_mlp(h, m + ".decoder_network.embed_node_fns.grid_nodes", L, 0, L, L, L, c.c_out, false, &cp, &h->m2g_dec))) return rc;

  for (DevMlp* em : {&h->g2m_edge, &h->m2g_edge}) {
    const std::string pth = (em == &h->g2m_edge) ? gn + ".update_edge_fns.grid2mesh.edge_fn"
                                                 : gn2 + ".update_edge_fns.mesh2grid.edge_fn";
    const auto& k1 = h->weights.at(pth + ".network.network.layers.0.kernel");   // [3L][L]
    const auto we = transpose_pad(k1, 3 * L, L, 0, L, L, L);
    const auto ws = transpose_pad(k1, 3 * L, L, L, L, L, L);
    const auto wr = transpose_pad(k1, 3 * L, L, 2 * L, L, L, L);
    if ((rc = dev_upload(h, &em->w1e_t, we))) return rc;
    if ((rc = dev_upload(h, &em->w1e_s, encode_s16(we, L, L)))) return rc;
    if ((rc = dev_upload(h, &em->w1snd_t, ws))) return rc;
    if ((rc = dev_upload(h, &em->w1snd_s, encode_s16(ws, L, L)))) return rc;
    if ((rc = dev_upload(h, &em->w1rcv_t, wr))) return rc;
    if ((rc = dev_upload(h, &em->w1rcv_s, encode_s16(wr, L, L)))) return rc;
    {
      const int lf = round_up(L, 64);           // K of the weight-streaming images (L is 128 / 256 / 512: lf == L)
      if ((rc = dev_upload(h, &em->w1e_f, encode_wf16(transpose_pad(k1, 3 * L, L, 0, L, lf, L), L, lf)))) return rc;
      if (h->f32_ws && (rc = dev_upload(h, &em->w1e_x, encode_wf32(transpose_pad(k1, 3 * L, L, 0, L, lf, L), L, lf)))) return rc;
      if ((rc = dev_upload(h, &em->w1snd_f, encode_wf16(ws, L, L)))) return rc;
      if ((rc = dev_upload(h, &em->w1rcv_f, encode_wf16(wr, L, L)))) return rc;
    }
  }
  h->layers.assign(c.num_layers, DevLayer());
  for (int i = 0; i < c.num_layers; ++i) {
    const std::string b = t + ".blocks." + std::to_string(i);
    DevLayer& ly = h->layers[i];
    std::vector<float> qkv((size_t)3 * D * D);
    int part = 0;
    for (const char* q : {"q", "k", "v"}) {
      const auto tt = transpose_pad(h->weights.at(b + ".attn_module." + q + "_proj.linear.kernel"), D, D, 0, D, D, D);
      std::copy(tt.begin(), tt.end(), qkv.begin() + (size_t)part * D * D);
      ++part;
    }
    if ((rc = dev_upload(h, &ly.wqkv_t, qkv))) return rc;
    if ((rc = dev_upload(h, &ly.wqkv_s, encode_s16(qkv, 3 * D, D)))) return rc;
    if ((rc = dev_upload(h, &ly.wqkv_f, encode_wf16(qkv, 3 * D, D)))) return rc;
    if (h->f32_ws && (rc = dev_upload(h, &ly.wqkv_x, encode_wf32(qkv, 3 * D, D)))) return rc;
    {
      const auto wo = transpose_pad(h->weights.at(b + ".attn_module.final_linear.kernel"), D, D, 0, D, D, D);
      if ((rc = dev_upload(h, &ly.wo_t, wo))) return rc;
      if ((rc = dev_upload(h, &ly.wo_s, encode_s16(wo, D, D)))) return rc;
      if ((rc = dev_upload(h, &ly.wo_f, encode_wf16(wo, D, D)))) return rc;
      if (h->f32_ws && (rc = dev_upload(h, &ly.wo_x, encode_wf32(wo, D, D)))) return rc;
    }
    if ((rc = dev_upload(h, &ly.bo, h->weights.at(b + ".attn_module.final_linear.bias")))) return rc;
    {
      const auto w1 = transpose_pad(h->weights.at(b + ".ffw_module.mlp.layers.0.kernel"), D, F, 0, D, D, F);
      if ((rc = dev_upload(h, &ly.w1_t, w1))) return rc;
      if ((rc = dev_upload(h, &ly.w1_s, encode_s16(w1, F, D)))) return rc;
      if ((rc = dev_upload(h, &ly.w1_f, encode_wf16(w1, F, D)))) return rc;
      if (h->f32_ws && (rc = dev_upload(h, &ly.w1_x, encode_wf32(w1, F, D)))) return rc;
    }
    if ((rc = dev_upload(h, &ly.b1, h->weights.at(b + ".ffw_module.mlp.layers.0.bias")))) return rc;
    {
      const auto w2 = transpose_pad(h->weights.at(b + ".ffw_module.mlp.layers.2.kernel"), F, D, 0, F, F, D);
      if ((rc = dev_upload(h, &ly.w2_t, w2))) return rc;
      if ((rc = dev_upload(h, &ly.w2_s, encode_s16(w2, D, F)))) return rc;
      if ((rc = dev_upload(h, &ly.w2_f, encode_wf16(w2, D, F)))) return rc;
      if (h->f32_ws && (rc = dev_upload(h, &ly.w2_x, encode_wf32(w2, D, F)))) return rc;
    }
    if ((rc = dev_upload(h, &ly.b2, h->weights.at(b + ".ffw_module.mlp.layers.2.bias")))) return rc;
    ly.cond_attn = cp.add(b + ".norm_cond_attn.conditional_linear_layer",
                          h->weights.at(b + ".norm_cond_attn.conditional_linear_layer.kernel"),
                          h->weights.at(b + ".norm_cond_attn.conditional_linear_layer.bias"), D);
    ly.cond_ffw = cp.add(b + ".norm_cond_ffw.conditional_linear_layer",
                         h->weights.at(b + ".norm_cond_ffw.conditional_linear_layer.kernel"),
                         h->weights.at(b + ".norm_cond_ffw.conditional_linear_layer.bias"), D);
  }
  h->cond_final = cp.add(t + ".final_norm_cond.conditional_linear_layer",
                         h->weights.at(t + ".final_norm_cond.conditional_linear_layer.kernel"),
                         h->weights.at(t + ".final_norm_cond.conditional_linear_layer.bias"), D);

  // all conditioning linears side by side: wc_all[16][total], bc_all[total] (+1 folded into scales)
  h->cond_total = cp.total;
  h->cond_sites = cp.sites;
  std::vector<float> wc((size_t)gc::kCondDim * cp.total), bc(cp.total);
  int off = 0;
  for (size_t li = 0; li < cp.kernels.size(); ++li) {
    const int cdim = cp.sizes[li];
    const auto& k = *cp.kernels[li];
    const auto& b = *cp.biases[li];
    for (int i = 0; i < gc::kCondDim; ++i)
      for (int j = 0; j < 2 * cdim; ++j) wc[(size_t)i * cp.total + off + j] = k[(size_t)i * 2 * cdim + j];
    for (int j = 0; j < 2 * cdim; ++j) bc[off + j] = b[j] + (j < cdim ? 1.0f : 0.0f);
    off += 2 * cdim;
  }
  if ((rc = dev_upload(h, &h->d_wc_all, wc))) return rc;
  if ((rc = dev_upload(h, &h->d_bc_all, bc))) return rc;
  if ((rc = dev_alloc(h, &h->d_cond, (size_t)c.batch * cp.total))) return rc;

  const int nf2 = 2 * c.noise_num_frequencies;
  if ((rc = dev_upload(h, &h->d_nw0t, transpose_pad(h->weights.at(nz + ".linear_0.kernel"), nf2, c.noise_hidden, 0, nf2, nf2, c.noise_hidden)))) return rc;
  if ((rc = dev_upload(h, &h->d_nb0, h->weights.at(nz + ".linear_0.bias")))) return rc;
  if ((rc = dev_upload(h, &h->d_nw1t, transpose_pad(h->weights.at(nz + ".linear_1.kernel"), c.noise_hidden, gc::kCondDim, 0, c.noise_hidden, c.noise_hidden, gc::kCondDim)))) return rc;
  if ((rc = dev_upload(h, &h->d_nb1, h->weights.at(nz + ".linear_1.bias")))) return rc;

  // f16x3 domain of the weights: a non-finite or > fp16-max weight cannot be split, so such a model
  // runs on the exact-f32 kernels only (the reference's f32 arithmetic has no such limit)
  h->weights_f16_unsafe = false;
  for (const auto& kv : h->weights)
    for (float w : kv.second)
      if (!(std::fabs(w) <= 65504.0f)) { h->weights_f16_unsafe = true; break; }
  if ((rc = compute_static_embeddings(h))) return rc;
  h->finalized = true;
  h->finalized_weights = true;
  if ((rc = build_embed_cache(h))) return rc;
  return GC_OK;
  });
}

int gc_denoise(gc_handle* h, const float* grid_feats, const float* sigma, float* out) {
  return guarded(h, [&]() -> int {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!grid_feats || !sigma || !out) return fail(h, GC_ERR_INVALID_ARGUMENT, "null argument");
  const gc_config& c = h->cfg;
  for (int b = 0; b < c.batch; ++b)
    if (!(sigma[b] > 0.f)) return fail(h, GC_ERR_INVALID_ARGUMENT, "noise levels must be > 0");
  GC_HIP(h, hipSetDevice(h->device));
  const size_t GB = (size_t)h->hg.G * c.batch;
  GC_HIP(h, hipMemcpyAsync(h->d_feats, grid_feats, GB * c.c_in * sizeof(float), hipMemcpyHostToDevice, h->stream));
  GC_HIP(h, hipMemcpyAsync(h->d_sigma, sigma, c.batch * sizeof(float), hipMemcpyHostToDevice, h->stream));
  if ((rc = launch(h, gc::KC_PACK, [&] {
         return gc::launch_pack_full(h->stream, h->d_grid_struct, h->d_feats, h->hg.G, c.batch, c.c_in,
                                     h->kp, h->d_xp);
       })))
    return rc;
  h->has_cond = false;  // d_xp no longer holds the sampler's conditioning
  if ((rc = forward(h, -1.0f))) return rc;
  if ((rc = guard_enqueue(h, h->d_y, GB * c.c_out))) return rc;
  GC_HIP(h, hipMemcpyAsync(out, h->d_y, GB * c.c_out * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  GC_HIP(h, hipStreamSynchronize(h->stream));
  if (use_f16(h) && guard_tripped(h)) {      // left the f16x3 domain: the same call on the exact-f32 kernels
    ++h->range_fallbacks;
    h->in_fallback = true;
    rc = forward(h, -1.0f);
    h->in_fallback = false;
    if (rc) return rc;
    GC_HIP(h, hipMemcpyAsync(out, h->d_y, GB * c.c_out * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    GC_HIP(h, hipStreamSynchronize(h->stream));
  }
  return GC_OK;
  });
}

int gc_set_noisy_slots(gc_handle* h, const int32_t* slots) {
  return guarded(h, [&]() -> int {
  if (!h) return GC_ERR_INVALID_ARGUMENT;
  if (!h->has_graph) return fail(h, GC_ERR_STATE, "gc_set_graph must be called first");
  if (!slots) return fail(h, GC_ERR_INVALID_ARGUMENT, "null argument");
  const gc_config& c = h->cfg;
  int rc;
  std::vector<char> seen(c.c_in, 0);
  for (int i = 0; i < c.c_out; ++i) {
    if (slots[i] < 0 || slots[i] >= c.c_in || seen[slots[i]])
      return fail(h, GC_ERR_INVALID_ARGUMENT, "noisy slots must be distinct columns of grid_feats");
    seen[slots[i]] = 1;
  }
  // the callers' samplers set the slots before every sample (gencast-flax-nnx_amd/sampler.py): the same slots again change
  // nothing -- no copy, and above all no rebuild of the split embedding images, which would drop the captured sample graphs
  if (h->has_slots && (int)h->h_slots.size() == c.c_out && std::equal(slots, slots + c.c_out, h->h_slots.begin()) &&
      (h->embed_cache_ready || !(h->embed_cache && h->finalized_weights && h->hidden_layers == 1 && h->mlp_ws)))
    return GC_OK;
  GC_HIP(h, hipSetDevice(h->device));
  if (h->guard_pending && (rc = resolve_guard(h))) return rc;   // a pending re-run must still see the old slots
  // on the handle's stream (it is non-blocking: a null-stream copy would not be ordered against a
  // sampler still running), then waited for, so `slots` is free on return
  GC_HIP(h, hipMemcpyAsync(h->d_slots, slots, c.c_out * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
  GC_HIP(h, hipStreamSynchronize(h->stream));
  h->has_slots = true;
  h->h_slots.assign(slots, slots + c.c_out);
  return build_embed_cache(h);                  // (no-op until gc_finalize has laid the weights out)
  });
}

int gc_commit_cond(gc_handle* h) {
  return guarded(h, [&]() -> int {
  int rc = check_ready(h);
  if (rc) return rc;
  const gc_config& c = h->cfg;
  GC_HIP(h, hipSetDevice(h->device));
  if (h->guard_pending && (rc = resolve_guard(h))) return rc;
  if ((rc = launch(h, gc::KC_PACK, [&] {
         return gc::launch_pack_full(h->stream, h->d_grid_struct, h->d_feats, h->hg.G, c.batch, c.c_in,
                                     h->kp, h->d_xp);
       })))
    return rc;
  h->has_cond = true;
  return GC_OK;
  });
}

int gc_upload_cond(gc_handle* h, const float* cond_feats) {
  return guarded(h, [&]() -> int {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!cond_feats) return fail(h, GC_ERR_INVALID_ARGUMENT, "null argument");
  GC_HIP(h, hipSetDevice(h->device));
  if (h->guard_pending && (rc = resolve_guard(h))) return rc;   // a pending re-run needs the conditioning it sampled with
  const size_t n = (size_t)h->hg.G * h->cfg.batch * h->cfg.c_in;
  if ((rc = staged_upload(h, h->pin_cond, h->d_feats, cond_feats, n))) return rc;
  return gc_commit_cond(h);
  });
}

int gc_upload_cond_dev(gc_handle* h, const void* cond_feats_dev) {
  return guarded(h, [&]() -> int {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!cond_feats_dev) return fail(h, GC_ERR_INVALID_ARGUMENT, "null argument");
  GC_HIP(h, hipSetDevice(h->device));
  if (h->guard_pending && (rc = resolve_guard(h))) return rc;
  const size_t n = (size_t)h->hg.G * h->cfg.batch * h->cfg.c_in;
  GC_HIP(h, hipMemcpyAsync(h->d_feats, cond_feats_dev, n * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
  return gc_commit_cond(h);
  });
}

int gc_cond_device_ptr(gc_handle* h, void** ptr, int64_t* nbytes) {
  return guarded(h, [&]() -> int {
  if (!h || !ptr || !nbytes) return GC_ERR_INVALID_ARGUMENT;
  if (!h->has_graph) return fail(h, GC_ERR_STATE, "gc_set_graph must be called first");
  *ptr = h->d_feats;
  *nbytes = (int64_t)h->hg.G * h->cfg.batch * h->cfg.c_in * (int64_t)sizeof(float);
  return GC_OK;
  });
}

int gc_upload_noise(gc_handle* h, const float* init_noise) {
  return guarded(h, [&]() -> int {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!init_noise) return fail(h, GC_ERR_INVALID_ARGUMENT, "null argument");
  GC_HIP(h, hipSetDevice(h->device));
  const size_t n = (size_t)h->hg.G * h->cfg.batch * h->cfg.c_out;
  protect_pending_noise(h);
  if ((rc = staged_upload(h, h->pin_noise, h->d_noise, init_noise, n))) return rc;
  h->has_noise = true;
  return GC_OK;
  });
}

int gc_sample_resident(gc_handle* h, const float* sigmas, int32_t n, int32_t skip_dead_call,
                       gc_sample_stats* stats) {
  return guarded(h, [&]() -> int {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!sigmas || n < 1) return fail(h, GC_ERR_INVALID_ARGUMENT, "need at least one noise level");
  if (!h->has_slots) return fail(h, GC_ERR_STATE, "gc_set_noisy_slots has not been called");
  if (!h->has_cond) return fail(h, GC_ERR_STATE, "no conditioning uploaded (gc_upload_cond)");
  if (!h->has_noise) return fail(h, GC_ERR_STATE, "no initial noise uploaded (gc_upload_noise)");
  for (int i = 0; i < n; ++i)
    if (!(sigmas[i] > 0.f) || !(sigmas[i + 1] >= 0.f) || !(sigmas[i + 1] < sigmas[i]))
      return fail(h, GC_ERR_INVALID_ARGUMENT, "sigmas must be positive and strictly descending (a trailing 0 is allowed)");
  GC_HIP(h, hipSetDevice(h->device));
  return run_sampler(h, sigmas, n, skip_dead_call, stats);
  });
}

int gc_download_sample(gc_handle* h, float* out) {
  return guarded(h, [&]() -> int {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!out) return fail(h, GC_ERR_INVALID_ARGUMENT, "null argument");
  GC_HIP(h, hipSetDevice(h->device));
  const size_t n = (size_t)h->hg.G * h->cfg.batch * h->cfg.c_out;
  if ((rc = resolve_guard(h))) return rc;
  GC_HIP(h, hipMemcpyAsync(out, h->d_sx, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  GC_HIP(h, hipStreamSynchronize(h->stream));
  return GC_OK;
  });
}

int gc_stash_sample(gc_handle* h) {
  return guarded(h, [&]() -> int {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!h->has_sample) return fail(h, GC_ERR_STATE, "no sample on the device (gc_sample_resident)");
  GC_HIP(h, hipSetDevice(h->device));
  if ((rc = resolve_guard(h))) return rc;        // the snapshot is of the CHECKED sample (exact-f32 re-run included)
  const size_t n = (size_t)h->hg.G * h->cfg.batch * h->cfg.c_out;
  if (!h->d_stash && (rc = dev_alloc(h, &h->d_stash, n))) return rc;
  if (!h->ev_stash) GC_HIP(h, hipEventCreateWithFlags(&h->ev_stash, hipEventDisableTiming));
  if (h->has_stash) GC_HIP(h, hipStreamSynchronize(h->stream2));   // an earlier snapshot's download has left the buffer
  GC_HIP(h, hipMemcpyAsync(h->d_stash, h->d_sx, n * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
  GC_HIP(h, hipEventRecord(h->ev_stash, h->stream));
  h->has_stash = true;
  return GC_OK;
  });
}

int gc_download_stash(gc_handle* h, float* out) {
  return guarded(h, [&]() -> int {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!out) return fail(h, GC_ERR_INVALID_ARGUMENT, "null argument");
  if (!h->has_stash) return fail(h, GC_ERR_STATE, "no snapshot on the device (gc_stash_sample)");
  GC_HIP(h, hipSetDevice(h->device));
  const size_t n = (size_t)h->hg.G * h->cfg.batch * h->cfg.c_out;
  // on the side stream, behind the snapshot copy only: whatever the main stream has been given since (the context
  // update, the next sample) keeps running while the host copy is in flight
  GC_HIP(h, hipStreamWaitEvent(h->stream2, h->ev_stash, 0));
  GC_HIP(h, hipMemcpyAsync(out, h->d_stash, n * sizeof(float), hipMemcpyDeviceToHost, h->stream2));
  GC_HIP(h, hipStreamSynchronize(h->stream2));
  return GC_OK;
  });
}

int gc_rollout_plan(gc_handle* h, const int32_t* kind, const int32_t* src, const int32_t* sidx, const float* a,
                    const float* b, int32_t n_forcing) {
  return guarded(h, [&]() -> int {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!kind || !src || !sidx || !a || !b || n_forcing < 0) return fail(h, GC_ERR_INVALID_ARGUMENT, "null argument");
  const gc_config& c = h->cfg;
  for (int i = 0; i < c.c_in; ++i) {
    const int k = kind[i];
    if (k < 0 || k > 4) return fail(h, GC_ERR_INVALID_ARGUMENT, "rollout plan: kind must be 0..4");
    if ((k == 1 || k == 2) && (src[i] < 0 || src[i] >= c.c_in))
      return fail(h, GC_ERR_INVALID_ARGUMENT, "rollout plan: src out of range");
    if ((k == 2 || k == 4) && (sidx[i] < 0 || sidx[i] >= c.c_out))
      return fail(h, GC_ERR_INVALID_ARGUMENT, "rollout plan: sample index out of range");
    if (k == 3 && (sidx[i] < 0 || sidx[i] >= n_forcing))
      return fail(h, GC_ERR_INVALID_ARGUMENT, "rollout plan: forcing index out of range");
  }
  GC_HIP(h, hipSetDevice(h->device));
  const size_t GB = (size_t)h->hg.G * c.batch;
  auto upi = [&](int** d, const int32_t* p) -> int {
    int r;
    if (!*d && (r = dev_alloc(h, d, (size_t)c.c_in))) return r;
    GC_HIP(h, hipMemcpyAsync(*d, p, c.c_in * sizeof(int), hipMemcpyHostToDevice, h->stream));
    return GC_OK;
  };
  auto upf = [&](float** d, const float* p) -> int {
    int r;
    if (!*d && (r = dev_alloc(h, d, (size_t)c.c_in))) return r;
    GC_HIP(h, hipMemcpyAsync(*d, p, c.c_in * sizeof(float), hipMemcpyHostToDevice, h->stream));
    return GC_OK;
  };
  if ((rc = upi(&h->d_ro_kind, kind)) || (rc = upi(&h->d_ro_src, src)) || (rc = upi(&h->d_ro_sidx, sidx)) ||
      (rc = upf(&h->d_ro_a, a)) || (rc = upf(&h->d_ro_b, b)))
    return rc;
  GC_HIP(h, hipStreamSynchronize(h->stream));   // the plan arrays are the caller's again
  if (!h->d_feats2 && (rc = dev_alloc(h, &h->d_feats2, GB * c.c_in))) return rc;
  if (n_forcing > h->ro_forc_cap) {            // (an earlier, smaller buffer stays owned by the handle)
    if ((rc = dev_alloc(h, &h->d_ro_forc, GB * n_forcing))) return rc;
    GC_HIP(h, hipEventSynchronize(h->ev_pin));
    if (h->pin_forc) GC_HIP(h, hipHostFree(h->pin_forc));
    h->pin_forc = nullptr;
    GC_HIP(h, hipHostMalloc((void**)&h->pin_forc, GB * n_forcing * sizeof(float), hipHostMallocDefault));
    h->ro_forc_cap = n_forcing;
  }
  h->ro_nforc = n_forcing;
  return GC_OK;
  });
}

int gc_rollout_advance(gc_handle* h, const float* forcings) {
  return guarded(h, [&]() -> int {
  int rc = check_ready(h);
  if (rc) return rc;
  if (h->ro_nforc < 0) return fail(h, GC_ERR_STATE, "gc_rollout_plan has not been called");
  if (!h->has_cond) return fail(h, GC_ERR_STATE, "no conditioning uploaded (gc_upload_cond)");
  if (!h->has_sample) return fail(h, GC_ERR_STATE, "no sample on the device (gc_sample_resident)");
  if (h->ro_nforc > 0 && !forcings) return fail(h, GC_ERR_INVALID_ARGUMENT, "the plan needs a forcings array");
  const gc_config& c = h->cfg;
  const size_t GB = (size_t)h->hg.G * c.batch;
  GC_HIP(h, hipSetDevice(h->device));
  if (h->guard_pending && (rc = resolve_guard(h))) return rc;   // the plan reads the sample
  if (h->ro_nforc > 0 && (rc = staged_upload(h, h->pin_forc, h->d_ro_forc, forcings, GB * h->ro_nforc))) return rc;
  if ((rc = launch(h, gc::KC_PACK, [&] {
         return gc::launch_rollout_advance(h->stream, h->d_feats, h->d_sx, h->d_ro_forc, h->d_ro_kind, h->d_ro_src,
                                           h->d_ro_sidx, h->d_ro_a, h->d_ro_b, (int)GB, c.c_in, c.c_out,
                                           h->ro_nforc, h->d_feats2);
       })))
    return rc;
  std::swap(h->d_feats, h->d_feats2);
  return gc_commit_cond(h);
  });
}

int gc_download_cond(gc_handle* h, float* out) {
  return guarded(h, [&]() -> int {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!out) return fail(h, GC_ERR_INVALID_ARGUMENT, "null argument");
  if (!h->has_cond) return fail(h, GC_ERR_STATE, "no conditioning uploaded (gc_upload_cond)");
  GC_HIP(h, hipSetDevice(h->device));
  const size_t n = (size_t)h->hg.G * h->cfg.batch * h->cfg.c_in;
  GC_HIP(h, hipMemcpyAsync(out, h->d_feats, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  GC_HIP(h, hipStreamSynchronize(h->stream));
  return GC_OK;
  });
}

int gc_sync(gc_handle* h) {
  return guarded(h, [&]() -> int {
  if (!h) return GC_ERR_INVALID_ARGUMENT;
  GC_HIP(h, hipSetDevice(h->device));
  return resolve_guard(h);
  });
}

int gc_sample(gc_handle* h, const float* cond_feats, const float* init_noise, const float* sigmas,
              int32_t n, int32_t skip_dead_call, float* out, gc_sample_stats* stats) {
  return guarded(h, [&]() -> int {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!cond_feats || !init_noise || !out) return fail(h, GC_ERR_INVALID_ARGUMENT, "null argument");
  if ((rc = gc_upload_cond(h, cond_feats))) return rc;
  if ((rc = gc_upload_noise(h, init_noise))) return rc;
  if ((rc = gc_sample_resident(h, sigmas, n, skip_dead_call, stats))) return rc;
  return gc_download_sample(h, out);
  });
}

int gc_num_kernel_classes(void) { return gc::KC_COUNT; }
const char* gc_kernel_class_name(int cls) { return gc::kernel_class_name(cls); }

int gc_profile_enable(gc_handle* h, int cls) {
  if (!h) return GC_ERR_INVALID_ARGUMENT;
  if (cls >= gc::KC_COUNT) return fail(h, GC_ERR_INVALID_ARGUMENT, "unknown kernel class");
  GC_HIP(h, hipSetDevice(h->device));
  GC_HIP(h, hipStreamSynchronize(h->stream));
  if (cls >= 0 && h->prof_events.empty()) {
    h->prof_events.resize(2 * 4096);
    for (auto& e : h->prof_events) GC_HIP(h, hipEventCreate(&e));
  }
  h->prof_cls = cls;
  h->prof_used = 0;
  return GC_OK;
}

int gc_profile_set_stride(gc_handle* h, int stride) {
  return guarded(h, [&]() -> int {
  if (!h) return GC_ERR_INVALID_ARGUMENT;
  if (stride < 1) return fail(h, GC_ERR_INVALID_ARGUMENT, "stride must be >= 1");
  h->prof_stride = stride;
  h->prof_seen = 0;
  return GC_OK;
  });
}

int gc_profile_read(gc_handle* h, int32_t* launches, float* total_ms) {
  return guarded(h, [&]() -> int {
  if (!h || !launches || !total_ms) return GC_ERR_INVALID_ARGUMENT;
  GC_HIP(h, hipSetDevice(h->device));
  GC_HIP(h, hipStreamSynchronize(h->stream));
  float tot = 0.f;
  for (size_t i = 0; i + 1 < h->prof_used; i += 2) {
    float ms = 0.f;
    GC_HIP(h, hipEventElapsedTime(&ms, h->prof_events[i], h->prof_events[i + 1]));
    tot += ms;
  }
  *launches = (int32_t)(h->prof_used / 2);
  *total_ms = tot;
  h->prof_used = 0;
  return GC_OK;
  });
}

int gc_algorithmic_work(gc_handle* h, double* flops, double* bytes) {
  return guarded(h, [&]() -> int {
  if (!h || !flops || !bytes) return GC_ERR_INVALID_ARGUMENT;
  if (!h->has_graph) return fail(h, GC_ERR_STATE, "gc_set_graph must be called first");
  const gc_config& c = h->cfg;
  const double B = c.batch, G = h->hg.G * B, M = h->hg.M * B, E1 = h->hg.E1 * B, E2 = h->hg.E2 * B;
  const double L = c.latent_size, D = c.d_model, F = c.ffw_hidden, NL = c.num_layers;
  const double nnz = (double)h->hg.khop_nnz * B;
  double f = 0;
  f += 2 * G * (3 + c.c_in) * L + 2 * G * L * L;   // grid embed
  f += 2 * E1 * 3 * L * L + 2 * E1 * L * L;        // g2m edge update
  f += 2 * M * 2 * L * L + 2 * M * L * L;          // g2m mesh update
  f += 2 * G * L * L * 2;                          // g2m grid update
  f += NL * (2 * M * D * 3 * D + 2 * M * D * D + 2 * M * D * F * 2 + 4 * nnz * D);
  f += 2 * E2 * 3 * L * L + 2 * E2 * L * L;        // m2g edge update
  f += 2 * G * 2 * L * L + 2 * G * L * L;          // m2g grid update
  f += 2 * G * L * L + 2 * G * L * c.c_out;        // decoder
  // hidden_layers >= 2: one more L x L Linear per extra hidden layer in each of the 10 MLPs (the three edge / mesh
  // embedders run once per set-up, not per call: not counted, as above)
  f += (h->hidden_layers - 1) * 2.0 * L * L * (G + E1 + M + G + E2 + G + G);
  double params = 0;
  for (const auto& kv : h->specs) {
    double n = 1;
    for (auto d : kv.second) n *= (double)d;
    params += n;
  }
  // weights once + input + output + grid latent w/r + per layer {x r/w x2, QKV w+r}
  *bytes = 4.0 * (params + G * (c.c_in + c.c_out) + 2 * G * L + NL * M * D * 10);
  *flops = f;
  return GC_OK;
  });
}

int gc_get_counter(gc_handle* h, const char* name, int64_t* value) {
  return guarded(h, [&]() -> int {
  if (!h) return GC_ERR_INVALID_ARGUMENT;
  if (!name || !value) return fail(h, GC_ERR_INVALID_ARGUMENT, "null argument");
  const std::string n(name);
  if (n == "range_fallbacks") *value = h->range_fallbacks;
  else if (n == "launches_per_call") *value = h->launches_last_call;
  else if (n == "weights_f16_unsafe") *value = h->weights_f16_unsafe ? 1 : 0;
  else if (n == "fp16_storage") *value = h->last_st16 ? 1 : 0;
  else if (n == "split_edge") *value = h->split_edge ? 1 : 0;
  else if (n == "attention_items") *value = h->last_att_items;
  else if (n == "m2g_fused_sum") *value = h->last_m2g_fused ? 1 : 0;
  else if (n == "embed_cache") *value = h->embed_cache_samples;
  else if (n == "graph_replays") *value = h->graph_replays;
  else if (n == "graph_captures") *value = h->graph_captures;
  else return fail(h, GC_ERR_INVALID_ARGUMENT, "unknown counter: " + n);
  return GC_OK;
  });
}

// ---- spherical white noise on the device + stochastic churn (include/gencast_hip.h) -----------------
int gc_noise_set_tables(gc_handle* h, int32_t n_lat, int32_t n_lon, int32_t lmax, const float* legendre,
                        const float* cos_table, const float* sin_table) {
  return guarded(h, [&]() -> int {
  if (!h) return GC_ERR_INVALID_ARGUMENT;
  if (!h->has_graph) return fail(h, GC_ERR_STATE, "gc_set_graph must be called first");
  if (!legendre || !cos_table || !sin_table) return fail(h, GC_ERR_INVALID_ARGUMENT, "null argument");
  if (n_lat < 2 || n_lon < 2 || lmax < 1 || (int64_t)n_lat * n_lon != h->hg.G)
    return fail(h, GC_ERR_INVALID_ARGUMENT, "n_lat * n_lon must equal the number of grid nodes");
  const int N = h->cfg.batch * h->cfg.c_out;
  // (no size limit: the Legendre step walks the latitudes in blocks of 192, the Fourier step stages wavenumber chunks
  //  of 64 columns through a fixed 64 KB of LDS -- samplers_utils.py:250-346 has none either)
  GC_HIP(h, hipSetDevice(h->device));
  int rc;
  const size_t L = (size_t)lmax;
  if ((rc = dev_upload(h, &h->d_nz_leg, std::vector<float>(legendre, legendre + L * n_lat * L)))) return rc;
  if ((rc = dev_upload(h, &h->d_nz_cos, std::vector<float>(cos_table, cos_table + (size_t)n_lon * L)))) return rc;
  if ((rc = dev_upload(h, &h->d_nz_sin, std::vector<float>(sin_table, sin_table + (size_t)n_lon * L)))) return rc;
  if ((rc = dev_alloc(h, &h->d_nz_coef, 2 * L * L * N + 4))) return rc;
  if ((rc = dev_alloc(h, &h->d_nz_f, 2 * L * n_lat * N))) return rc;
  h->nz_L = lmax; h->nz_lat = n_lat; h->nz_lon = n_lon;
  return GC_OK;
  });
}

int gc_noise_seed(gc_handle* h, uint64_t seed, uint64_t stream) {
  return guarded(h, [&]() -> int {
  if (!h) return GC_ERR_INVALID_ARGUMENT;
  if (h->guard_pending) {                      // a pending re-run draws its churn noise from the old key
    GC_HIP(h, hipSetDevice(h->device));
    int rc = resolve_guard(h);
    if (rc) return rc;
  }
  h->nz_key = seed;
  h->nz_stream = stream;
  return GC_OK;
  });
}

int gc_noise_draw(gc_handle* h) {
  return guarded(h, [&]() -> int {
  int rc = check_ready(h);
  if (rc) return rc;
  GC_HIP(h, hipSetDevice(h->device));
  protect_pending_noise(h);
  if ((rc = noise_field(h, nullptr, 1.0f, h->d_noise))) return rc;
  h->has_noise = true;
  return GC_OK;
  });
}

int gc_download_noise(gc_handle* h, float* out) {
  return guarded(h, [&]() -> int {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!out) return fail(h, GC_ERR_INVALID_ARGUMENT, "null argument");
  if (!h->has_noise) return fail(h, GC_ERR_STATE, "no initial noise on the device");
  GC_HIP(h, hipSetDevice(h->device));
  const size_t n = (size_t)h->hg.G * h->cfg.batch * h->cfg.c_out;
  GC_HIP(h, hipMemcpyAsync(out, h->d_noise, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  GC_HIP(h, hipStreamSynchronize(h->stream));
  return GC_OK;
  });
}

int gc_set_churn(gc_handle* h, const float* rates, int32_t n, float noise_level_inflation_factor) {
  return guarded(h, [&]() -> int {
  if (!h) return GC_ERR_INVALID_ARGUMENT;
  if (n < 0 || (n > 0 && !rates)) return fail(h, GC_ERR_INVALID_ARGUMENT, "null argument");
  bool any = false;
  for (int i = 0; i < n; ++i) {
    if (!(rates[i] >= 0.f)) return fail(h, GC_ERR_INVALID_ARGUMENT, "churn rates must be >= 0");
    any = any || rates[i] > 0.f;
  }
  if (any && h->nz_L == 0) return fail(h, GC_ERR_STATE, "stochastic churn needs the noise tables (gc_noise_set_tables)");
  if (h->guard_pending) {                      // a pending re-run repeats the schedule it sampled with
    GC_HIP(h, hipSetDevice(h->device));
    int rc = resolve_guard(h);
    if (rc) return rc;
  }
  if (any) h->churn_rates.assign(rates, rates + n);
  else h->churn_rates.clear();
  h->churn_inflation = noise_level_inflation_factor;
  return GC_OK;
  });
}

// ---- ensemble exchange over RCCL / xGMI (include/gencast_hip.h) -------------------------------------
int gc_comm_unique_id(void* id_out) {
  return guarded(nullptr, [&]() -> int {
  if (!id_out) { g_create_error = "null argument"; return GC_ERR_INVALID_ARGUMENT; }
  Rccl& r = rccl();
  if (!r.error.empty()) { g_create_error = r.error; return GC_ERR_COMM; }
  ncclUniqueId id;
  const ncclResult_t rc = r.GetUniqueId(&id);
  if (rc != ncclSuccess) { g_create_error = std::string("ncclGetUniqueId: ") + r.GetErrorString(rc); return GC_ERR_COMM; }
  static_assert(sizeof(id) == GC_COMM_ID_BYTES, "ncclUniqueId size");
  std::memcpy(id_out, &id, sizeof(id));
  return GC_OK;
  });
}

int gc_comm_init(gc_handle* h, const void* id, int32_t rank, int32_t world_size) {
  return guarded(h, [&]() -> int {
  if (!h) return GC_ERR_INVALID_ARGUMENT;
  if (!id || world_size < 1 || rank < 0 || rank >= world_size) return fail(h, GC_ERR_INVALID_ARGUMENT, "bad rank / world size");
  if (h->comm) return fail(h, GC_ERR_STATE, "communicator already initialised on this handle");
  Rccl& r = rccl();
  if (!r.error.empty()) return fail(h, GC_ERR_COMM, r.error);
  GC_HIP(h, hipSetDevice(h->device));
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  ncclComm_t comm = nullptr;                 // assigned to the handle only once it is valid
  GC_NCCL(h, r.CommInitRank(&comm, world_size, uid, rank));
  h->comm = comm;
  h->comm_rank = rank;
  h->comm_world = world_size;
  if (!h->d_comm_scalar) {
    int rc = dev_alloc(h, &h->d_comm_scalar, 1);
    if (rc) return rc;
  }
  return GC_OK;
  });
}

int gc_comm_info(gc_handle* h, int32_t* num_ranks, int32_t* rank) {
  return guarded(h, [&]() -> int {
  if (!h) return GC_ERR_INVALID_ARGUMENT;
  if (!num_ranks || !rank) return fail(h, GC_ERR_INVALID_ARGUMENT, "null argument");
  *num_ranks = 0;
  *rank = -1;
  if (!h->comm) return GC_OK;                  // no communicator: 0 ranks
  int n = 0, r = -1;
  GC_NCCL(h, rccl().CommCount(h->comm, &n));
  GC_NCCL(h, rccl().CommUserRank(h->comm, &r));
  *num_ranks = n;
  *rank = r;
  return GC_OK;
  });
}

int gc_comm_broadcast_cond(gc_handle* h, int32_t root) {
  return guarded(h, [&]() -> int {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!h->comm) return fail(h, GC_ERR_STATE, "gc_comm_init has not been called");
  if (root < 0 || root >= h->comm_world) return fail(h, GC_ERR_INVALID_ARGUMENT, "root out of range");
  GC_HIP(h, hipSetDevice(h->device));
  if (h->guard_pending && (rc = resolve_guard(h))) return rc;
  const size_t n = (size_t)h->hg.G * h->cfg.batch * h->cfg.c_in;
  // in place on the resident conditioning, ordered on the handle's stream behind any pending
  // upload (root) and in front of the re-pack below
  GC_NCCL(h, rccl().Broadcast(h->d_feats, h->d_feats, n, ncclFloat32, root, h->comm, h->stream));
  return gc_commit_cond(h);
  });
}

int gc_comm_allreduce_max(gc_handle* h, double* value) {
  return guarded(h, [&]() -> int {
  if (!h) return GC_ERR_INVALID_ARGUMENT;
  if (!value) return fail(h, GC_ERR_INVALID_ARGUMENT, "null argument");
  if (!h->comm) return fail(h, GC_ERR_STATE, "gc_comm_init has not been called");
  GC_HIP(h, hipSetDevice(h->device));
  GC_HIP(h, hipMemcpyAsync(h->d_comm_scalar, value, sizeof(double), hipMemcpyHostToDevice, h->stream));
  GC_NCCL(h, rccl().AllReduce(h->d_comm_scalar, h->d_comm_scalar, 1, ncclFloat64, ncclMax, h->comm, h->stream));
  GC_HIP(h, hipMemcpyAsync(value, h->d_comm_scalar, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  GC_HIP(h, hipStreamSynchronize(h->stream));
  return GC_OK;
  });
}

int gc_comm_destroy(gc_handle* h) {
  return guarded(h, [&]() -> int {
  if (!h) return GC_ERR_INVALID_ARGUMENT;
  if (!h->comm) return GC_OK;
  GC_HIP(h, hipSetDevice(h->device));
  GC_HIP(h, hipStreamSynchronize(h->stream));
  GC_NCCL(h, rccl().CommDestroy(h->comm));
  h->comm = nullptr;
  h->comm_world = 1;
  h->comm_rank = 0;
  return GC_OK;
  });
}

// ---- debug: fetch an intermediate buffer of the last forward (include/gencast_hip_debug.h) ----
int gc_debug_fetch(gc_handle* h, const char* name, float* out, int64_t capacity, int64_t* rows,
                   int64_t* cols) {
  return guarded(h, [&]() -> int {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!name || !rows || !cols) return fail(h, GC_ERR_INVALID_ARGUMENT, "null argument");
  const gc_config& c = h->cfg;
  const gc::HostGraph& g = h->hg;
  const int64_t B = c.batch, L = c.latent_size;
  // act: an activation buffer -- a _Float16 array when the last forward ran with physical fp16 storage
  struct Ent { const char* n; const float* p; int64_t items; int64_t w; int mesh; int batched; int act; };
  const Ent table[] = {
      {"cond", h->d_condvec, 1, gc::kCondDim, 0, 1, 0},
      {"g0", h->d_g0, g.G, L, 0, 1, 1},        {"m0", h->d_m0, g.M, L, 1, 1, 1},
      {"e1", h->d_e1, g.E1, L, 0, 1, 1},       {"agg1", h->d_agg1, g.M, L, 1, 1, 1},
      {"m1", h->d_x, g.M, L, 1, 1, 1},         {"x", h->d_x, g.M, L, 1, 1, 1},
      {"g1", h->d_g1, g.G, L, 0, 1, 1},        {"qkv", h->d_qkv, g.M, 3 * L, 1, 1, 0},
      {"att", h->d_att, g.M, L, 1, 1, 1},      {"m2", h->d_m2, g.M, L, 1, 1, 1},
      {"h", h->d_h, g.M, L, 1, 1, 1},
      {"f1", h->d_f1, g.E2, L, 0, 1, 1},       {"agg2", h->d_agg2, g.G, L, 0, 1, 1},
      {"g2", h->d_g2, g.G, L, 0, 1, 1},        {"y", h->d_y, g.G, c.c_out, 0, 1, 0},
      {"m0_hat", h->d_m0_hat, g.M, L, 1, 0, 0}, {"e0_hat", h->d_e0_hat, g.E1, L, 0, 0, 0},
      {"f0_hat", h->d_f0_hat, g.E2, L, 0, 0, 0},
  };
  if (!std::strncmp(name, "cond:", 5)) {
    // the [scale | offset] vectors one conditioning linear produced in the last gc_denoise ("+1" folded into the scale)
    auto it = h->cond_sites.find(name + 5);
    if (it == h->cond_sites.end()) return fail(h, GC_ERR_INVALID_ARGUMENT, std::string("unknown conditioning site: ") + (name + 5));
    const int off = it->second.first, w = it->second.second;
    *rows = B;
    *cols = 2 * w;
    if (!out) return GC_OK;
    if (capacity < *rows * *cols) return fail(h, GC_ERR_INVALID_ARGUMENT, "output buffer too small");
    GC_HIP(h, hipSetDevice(h->device));
    GC_HIP(h, hipStreamSynchronize(h->stream));
    for (int64_t b = 0; b < B; ++b)
      GC_HIP(h, hipMemcpy(out + b * 2 * w, h->d_cond + (size_t)b * h->cond_total + off, (size_t)2 * w * sizeof(float), hipMemcpyDeviceToHost));
    return GC_OK;
  }
  for (const Ent& e : table) {
    if (std::strcmp(e.n, name)) continue;
    const int64_t bb = e.batched ? B : 1;
    *rows = e.items * bb;
    *cols = e.w;
    if (!out) return GC_OK;
    if (capacity < *rows * *cols) return fail(h, GC_ERR_INVALID_ARGUMENT, "output buffer too small");
    GC_HIP(h, hipSetDevice(h->device));
    if (!std::strcmp(name, "f1") && h->last_m2g_fused) {
      // the last forward summed the updated edges inside the edge MLP and never stored them: run that MLP once more,
      // unfused, on the inputs the forward left behind (m2 / g1 or their per-node products, the call's conditioning)
      if (!h->cond_cur) return fail(h, GC_ERR_STATE, "no forward has run yet");
      if ((rc = run_m2g_edge(h, h->cond_cur, false))) return rc;
    }
    GC_HIP(h, hipStreamSynchronize(h->stream));
    std::vector<float> tmp((size_t)(*rows * *cols));
    if (e.act && h->last_st16) {                 // halfs in HBM: widen
      std::vector<uint16_t> hv(tmp.size());
      GC_HIP(h, hipMemcpy(hv.data(), e.p, hv.size() * sizeof(uint16_t), hipMemcpyDeviceToHost));
      for (size_t i = 0; i < hv.size(); ++i) tmp[i] = f16_bits_to_f32(hv[i]);
    } else {
      GC_HIP(h, hipMemcpy(tmp.data(), e.p, tmp.size() * sizeof(float), hipMemcpyDeviceToHost));
    }
    if (!std::strcmp(name, "qkv") && h->kv16_live && h->last_st16) {
      // physical fp16 storage: q is a _Float16 array [rows][D] at d_qkv; k, v are the hi planes of kv16 (lo unwritten)
      const size_t D = (size_t)c.d_model, nrows = (size_t)*rows;
      std::vector<uint16_t> q16(nrows * D), pl(nrows * 4 * D);
      GC_HIP(h, hipMemcpy(q16.data(), h->d_qkv, q16.size() * sizeof(uint16_t), hipMemcpyDeviceToHost));
      GC_HIP(h, hipMemcpy(pl.data(), h->d_kv16, pl.size() * sizeof(uint16_t), hipMemcpyDeviceToHost));
      for (size_t r0 = 0; r0 < nrows; ++r0)
        for (size_t d0 = 0; d0 < D; ++d0) {
          tmp[r0 * 3 * D + d0] = f16_bits_to_f32(q16[r0 * D + d0]);
          tmp[r0 * 3 * D + D + d0] = f16_bits_to_f32(pl[r0 * 4 * D + d0]);
          tmp[r0 * 3 * D + 2 * D + d0] = f16_bits_to_f32(pl[r0 * 4 * D + 2 * D + d0]);
        }
    } else if (!std::strcmp(name, "qkv") && h->kv16_live) {
      // the projection wrote k and v as fp16 hi / lo planes only: value = hi + lo / 2048
      const size_t D = (size_t)c.d_model, nrows = (size_t)*rows;
      std::vector<uint16_t> pl(nrows * 4 * D);
      GC_HIP(h, hipMemcpy(pl.data(), h->d_kv16, pl.size() * sizeof(uint16_t), hipMemcpyDeviceToHost));
      for (size_t r0 = 0; r0 < nrows; ++r0)
        for (int w = 0; w < 2; ++w)
          for (size_t d0 = 0; d0 < D; ++d0)
            tmp[r0 * 3 * D + (w + 1) * D + d0] = f16_bits_to_f32(pl[r0 * 4 * D + w * 2 * D + d0]) +
                                                 f16_bits_to_f32(pl[r0 * 4 * D + w * 2 * D + D + d0]) / 2048.0f;
    }
    if (e.mesh) {  // back to the caller's mesh numbering
      for (int64_t ni = 0; ni < e.items; ++ni)
        std::memcpy(out + (size_t)g.perm[ni] * bb * e.w, tmp.data() + (size_t)ni * bb * e.w,
                    (size_t)(bb * e.w) * sizeof(float));
    } else if (!std::strcmp(name, "f1") || !std::strcmp(name, "f0_hat")) {  // back to the caller's mesh2grid edge order
      for (int64_t ei = 0; ei < e.items; ++ei)
        std::memcpy(out + (size_t)g.m2g_order[ei] * bb * e.w, tmp.data() + (size_t)ei * bb * e.w,
                    (size_t)(bb * e.w) * sizeof(float));
    } else {
      std::memcpy(out, tmp.data(), tmp.size() * sizeof(float));
    }
    return GC_OK;
  }
  return fail(h, GC_ERR_INVALID_ARGUMENT, std::string("unknown debug buffer: ") + name);
  });
}

#ifdef GC_STAMPS
// Diagnostic build only (csrc/build.sh stamps): per-wave stamps of the LAST attention launch.
int gc_debug_attention_stamps(gc_handle* h, unsigned long long* out, int64_t words) {
  if (!h || (!out && words >= 0)) return GC_ERR_INVALID_ARGUMENT;
  static unsigned long long* dbuf = nullptr;
  static int64_t cap = 0;
  if (words < 0) {                               // arm: allocate -words and install
    if (cap < -words) { (void)hipFree(dbuf); if (hipMalloc((void**)&dbuf, (size_t)(-words) * 8) != hipSuccess) return GC_ERR_HIP; cap = -words; }
    (void)hipMemset(dbuf, 0, (size_t)cap * 8);
    return gc::set_attention_stamp_buffer(dbuf) == hipSuccess ? GC_OK : GC_ERR_HIP;
  }
  (void)hipStreamSynchronize(h->stream);
  return hipMemcpy(out, dbuf, (size_t)std::min(words, cap) * 8, hipMemcpyDeviceToHost) == hipSuccess ? GC_OK : GC_ERR_HIP;
}
#endif

int gc_debug_set_layer_limit(gc_handle* h, int32_t num_layers) {
  return guarded(h, [&]() -> int {
  if (!h) return GC_ERR_INVALID_ARGUMENT;
  h->debug_layer_limit = num_layers;
  return GC_OK;
  });
}

int gc_debug_set_stop(gc_handle* h, int32_t layer, int32_t phase) {
  return guarded(h, [&]() -> int {
  if (!h) return GC_ERR_INVALID_ARGUMENT;
  if (layer >= 0 && (phase < 0 || phase > 2)) return fail(h, GC_ERR_INVALID_ARGUMENT, "phase must be 0, 1 or 2");
  h->debug_stop_layer = layer;
  h->debug_stop_phase = layer >= 0 ? phase : -1;
  return GC_OK;
  });
}

int gc_debug_mesh_permutation(gc_handle* h, int32_t* perm_out) {
  return guarded(h, [&]() -> int {
  if (!h || !perm_out) return GC_ERR_INVALID_ARGUMENT;
  if (!h->has_graph) return fail(h, GC_ERR_STATE, "gc_set_graph must be called first");
  for (int i = 0; i < h->hg.M; ++i) perm_out[i] = h->hg.perm[i];
  return GC_OK;
  });
}

int gc_debug_attention_stats(gc_handle* h, int64_t* n_tiles, int64_t* n_chunks, int64_t* khop_nnz) {
  return guarded(h, [&]() -> int {
  if (!h || !n_tiles || !n_chunks || !khop_nnz) return GC_ERR_INVALID_ARGUMENT;
  if (!h->has_graph) return fail(h, GC_ERR_STATE, "gc_set_graph must be called first");
  *n_tiles = h->hg.n_tiles;
  *n_chunks = h->hg.tile_chunk_start.back();
  *khop_nnz = h->hg.khop_nnz;
  return GC_OK;
  });
}

}  // extern "C"
