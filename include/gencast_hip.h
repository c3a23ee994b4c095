/*
 * gencast_hip.h -- C ABI of libgencast_hip.so: the MI355X-native GenCast denoiser
 * forward + DPM-Solver++2S sampling path.
 *
 * The reference (fgiral000/gencast-flax-nnx) has no FFI layer: this path sits
 * behind three Python call contracts.  Each entry point below
 * names the reference interface it replaces (file:line into the reference).
 *
 * Conventions
 *   - every function returns an int status: 0 = OK, non-zero = gc_status code;
 *     the text of the last error on a handle is gc_last_error(h)
 *     (gc_last_error(NULL) = last error of a failed gc_create on this thread);
 *   - no C++ exceptions cross this boundary; no torch / framework types;
 *   - the CALLER owns every host buffer passed in; the LIBRARY owns all device
 *     memory, inside the opaque handle;
 *   - one handle = one GPU = one HIP stream; handles are independent, so N host
 *     threads (or N processes) may drive N GPUs concurrently.  A handle is not
 *     re-entrant;
 *   - all tensors are float32, row-major, node-major / batch-minor:
 *     [nodes, batch, channels] exactly like the reference's flat layout
 *     (gencast/denoiser.py:770-807); index arrays are int32;
 *   - there is NO CPU fallback: without a usable HIP device gc_create fails with
 *     GC_ERR_NO_DEVICE.
 */
#ifndef GENCAST_HIP_H_
#define GENCAST_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GC_ABI_VERSION 1

typedef struct gc_handle gc_handle;

typedef enum gc_status {
  GC_OK = 0,
  GC_ERR_INVALID_ARGUMENT = 1, /* bad dims / null pointer / unknown name (Python side raises ValueError) */
  GC_ERR_NO_DEVICE = 2,        /* no HIP device, or device_id out of range */
  GC_ERR_HIP = 3,              /* a HIP runtime call failed; text in gc_last_error */
  GC_ERR_STATE = 4,            /* call order violated (e.g. gc_denoise before gc_finalize) */
  GC_ERR_UNSUPPORTED = 5,      /* valid request outside what the kernels are built for */
  GC_ERR_INTERNAL = 6,         /* a C++ exception (e.g. out of host memory) was caught at the boundary */
  GC_ERR_COMM = 7              /* RCCL could not be loaded, or an RCCL call failed */
} gc_status;

/*
 * Model dimensions.  Mirrors DenoiserArchitectureConfig / SparseTransformerConfig /
 * NoiseEncoderConfig (gencast/denoiser.py:47-139) plus the data widths the
 * reference infers lazily on first call (gencast/denoiser.py:343-352).
 */
typedef struct gc_config {
  int32_t latent_size;      /* GNN latent = MLP hidden width (latent_size, hidden_layers = 1) */
  int32_t d_model;          /* transformer width; must equal latent_size */
  int32_t num_heads;
  int32_t ffw_hidden;
  int32_t num_layers;
  int32_t c_in;             /* stacked input + forcing channels per grid node (262 for the nano task) */
  int32_t c_out;            /* predicted channels per grid node (82) */
  int32_t batch;            /* B: batch / members evaluated together on this GPU */
  int32_t noise_num_frequencies; /* NoiseEncoderConfig.num_frequencies (32) */
  int32_t noise_hidden;          /* NoiseEncoderConfig.output_sizes[0] (32); [1] is fixed at 16 */
  float   noise_base_period;     /* NoiseEncoderConfig.base_period (16.0); apply_log_first = True */
} gc_config;

typedef struct gc_sample_stats {
  int32_t denoiser_calls;   /* network evaluations executed (39, or 40 with the dead call) */
  float   device_ms;        /* HIP-event time of the whole loop on the handle's stream */
} gc_sample_stats;

/* Library / build info; callable without a GPU. */
int         gc_abi_version(void);
const char* gc_build_info(void);               /* "... src:<16 hex>": hash of the csrc/ sources the library was built from (build.sh) */
int         gc_device_count(void);             /* 0 when no HIP device is visible */
/* PCI bus id ("0000:c1:00.0") of visible device `device_id` into out[cap] (cap >= 16): lets the ranks of one
 * launch check, before any collective, that no two of them sit on the same GPU (RCCL refuses that, and the rank
 * that entered ncclCommInitRank first would never leave it).  GC_ERR_INVALID_ARGUMENT for a bad index / buffer. */
int         gc_device_pci_bus_id(int32_t device_id, char* out, int64_t cap);
const char* gc_last_error(const gc_handle* h);

/*
 * Replaces: Denoiser.__init__ / DenoiserArchitecture.__init__
 * (gencast/denoiser.py:153-170, 231-301): allocates the handle on `device_id`
 * with its own stream.
 */
int gc_create(const gc_config* cfg, int device_id, gc_handle** out);
void gc_destroy(gc_handle* h);

/*
 * Runtime options (string key / value), callable any time after gc_create:
 *   "precision" = "f16x3" (default) | "f32"
 *       f16x3: every GEMM-shaped product runs as 3 fp16 MFMAs on operands split into
 *              hi + lo/2048 (22 significant bits; measured parity identical to f32);
 *       f32:   v_mfma_f32_32x32x2_f32 (exact f32 FMA chains), about 1.8x slower end to end.
 *   The environment variable GC_PRECISION sets the default for new handles.
 *   f16x3 domain: every GEMM operand must be finite with |x| <= 65504.  Nothing is clamped: an
 *   operand outside that range (an un-normalised field, NaN, Inf) poisons the result with NaN / Inf,
 *   the library checks the result of every call on the device and RE-RUNS a poisoned call on the
 *   exact-f32 kernels, whose treatment of such inputs is the reference's (f32 arithmetic, NaN / Inf
 *   propagate).  gc_get_counter("range_fallbacks") counts those re-runs.  For the resident
 *   (asynchronous) sampler the check is resolved by the next gc_download_sample / gc_sync /
 *   gc_rollout_advance.  Weights beyond the domain switch the handle to f32 kernels for good.
 *   "features" = "f32" (default) | "f16"   -- BASELINE.json configs[4], "fp16 node features":
 *       every activation is rounded to fp16 (round to nearest even) where it is produced -- grid
 *       input features, MLP hidden and output, LayerNorm + conditioning outputs, residual sums,
 *       q / k / v, softmax weights, attention output, segment sums -- while weights, conditioning
 *       vectors, GEMM accumulation, LayerNorm statistics, the softmax max / sum and the segment-sum
 *       accumulation stay float32: the points where the reference upcasts
 *       (gencast/sparse_transformer_utils.py:42-76, common/deep_typed_graph_net.py:396-403).  Attention
 *       then runs one fp16 MFMA per product instead of three, every other product two.  With precision
 *       f16x3 the activations are also STORED as 2-byte fp16 arrays in HBM (grid / mesh / edge latents, the
 *       residual stream, q / k / v, attention output, FFW hidden activation, segment sums: half the
 *       activation bytes of every kernel; split-K slabs, attention partials, the statically embedded
 *       latents and the network output stay float32); with precision f32, and in the exact-f32 re-run
 *       of the domain guard, the same values live in float32 containers.  Tensors at this boundary
 *       stay float32; parity tolerance vs the oracle in the same mode: DESIGN.md 3b.
 *   "graphs" = "on" (default) | "off"   -- HIP-graph replay of the sampler.  The reference's sampler is ONE compiled
 *       program (the jax.lax.fori_loop of gencast/dpm_solver_plus_plus_2s.py:157-158); here a sample is ~3 500
 *       kernel launches whose sequence depends only on (noise levels, skip_dead_call, precision, features), so the
 *       second gc_sample* call with one signature is captured into a hipGraph and later ones are a single
 *       hipGraphLaunch (host cost per sample: milliseconds -> tens of microseconds; same kernels, same arguments,
 *       same order: bit-identical samples).  Samples with stochastic churn, per-class profiling or the debug stops
 *       are always enqueued eagerly.  GC_TUNE_GRAPH=0 sets the default to off.
 *   "hidden_layers" = "1" (default) .. "4"   -- DenoiserArchitectureConfig.hidden_layers (gencast/denoiser.py:108,135;
 *       common/mlp.py:157-199): hidden layers of every MLP of the grid2mesh / mesh2grid GNNs.  It fixes the parameter
 *       names (`...network.network.layers.{0,2,..,2 n}`), so it must be set before the first gc_load_weight
 *       (GC_ERR_STATE afterwards).  n >= 2 runs every MLP as a chain of n fused launches, in both feature modes (with
 *       "f16" every hidden activation is an fp16 rounding point, the hand-over between the launches a float32 container
 *       of fp16 values); the reference trains with 1 (training/train_helpers.py:137), the tuned one-launch-per-MLP path.
 *   "grid2mesh_aggregate_normalization" = "none" (default) | "<positive constant>"   -- the summed grid2mesh edge
 *       messages of every mesh node are divided by it (common/deep_typed_graph_net.py:396-410; gencast/denoiser.py:123,138).
 */
int gc_set_option(gc_handle* h, const char* key, const char* value);

/*
 * Replaces: DenoiserArchitecture._maybe_init graph construction
 * (gencast/denoiser.py:343-363, 443-600) and Transformer.__init__'s mask set-up
 * (gencast/sparse_transformer.py:555-567).  The caller supplies the static graph
 * (built by gencast-flax-nnx_amd/geometry.py, or any other source):
 *   g2m edges   grid -> mesh,  E1 of them   (senders index grid nodes)
 *   m2g edges   mesh -> grid,  E2 of them   (senders index mesh nodes)
 *   khop CSR    row i = mesh nodes node i attends to (must contain i)
 *   *_struct    structural features: nodes (cos theta, cos phi, sin phi) [N,3],
 *               edges (|d|, d)/max|d| [E,4]  (common/model_utils.py:445-495)
 *   mesh_xyz    [M,3] unit vectors, used only to choose a cache-friendly internal
 *               node order; may be NULL.
 * Mesh node numbering is the caller's; the library renumbers internally.
 */
int gc_set_graph(gc_handle* h,
                 int32_t num_grid_nodes, int32_t num_mesh_nodes,
                 int32_t num_g2m_edges, const int32_t* g2m_senders, const int32_t* g2m_receivers,
                 int32_t num_m2g_edges, const int32_t* m2g_senders, const int32_t* m2g_receivers,
                 const int32_t* khop_rowptr, const int32_t* khop_cols,
                 const float* grid_struct, const float* mesh_struct,
                 const float* g2m_edge_struct, const float* m2g_edge_struct,
                 const float* mesh_xyz);

/*
 * Replaces: nnx.update(model, state) of a restored checkpoint
 * (training/evaluation.py:119-187).  `name` is the Flax-NNX attribute path of the
 * parameter (gencast-flax-nnx_amd/weights.py lists all of them),
 * `data` is row-major float32 with flax's kernel orientation (in, out).
 * Unknown names fail with GC_ERR_INVALID_ARGUMENT, except the reference's dead
 * mesh2grid mesh-node update, which is accepted and ignored.
 */
int gc_load_weight(gc_handle* h, const char* name, const float* data,
                   const int64_t* shape, int32_t ndim);

/* Number of parameters still missing (0 = ready for gc_finalize). */
int gc_missing_weights(gc_handle* h, int32_t* count);

/*
 * Checks that graph + every weight are present, lays weights out for the kernels
 * and pre-computes the embeddings whose inputs are static.
 * Must be called once before gc_denoise / gc_sample; call it again after
 * re-loading weights.
 */
int gc_finalize(gc_handle* h);

/*
 * Replaces: Denoiser.__call__ (gencast/denoisers_base.py:28-52;
 * gencast/denoiser.py:172-202 -> DenoiserArchitecture.__call__ :303-341), at the
 * flat-array level: raw network output F(X; sigma).
 *   grid_feats  [G, B, c_in]  inputs ++ forcings (noisy targets already scaled by c_in(sigma))
 *   sigma       [B]           noise levels (> 0)
 *   out         [G, B, c_out]
 * Host pointers; the call returns after the result has been copied back.
 */
int gc_denoise(gc_handle* h, const float* grid_feats, const float* sigma, float* out);

/*
 * Which columns of grid_feats hold the noisy targets, in output-channel order
 * (reference: `forcings.assign(noisy_targets)` + sorted-name stacking,
 * gencast/denoiser.py:184,770-807).  Needed by gc_sample / gc_sample_resident.
 */
int gc_set_noisy_slots(gc_handle* h, const int32_t* slots /* [c_out] */);

/*
 * Replaces: Sampler.__call__ of DPM-Solver++2S
 * (gencast/samplers_base.py:22-44; gencast/dpm_solver_plus_plus_2s.py:47-177,
 * preconditioning :181-205); stochastic churn when gc_set_churn installed a schedule.
 *   cond_feats  [G, B, c_in]   inputs ++ forcings; the noisy-slot columns are ignored
 *   init_noise  [G, B, c_out]  unit-variance noise; x0 = init_noise * sigmas[0]
 *   sigmas      [n + 1]        descending noise levels ending in 0 (samplers_utils.py:395-412)
 *   skip_dead_call  1 = omit the last step's mid-point evaluation whose result the
 *               reference discards (dpm_solver_plus_plus_2s.py:148-153); same output
 *   out         [G, B, c_out]  the sample
 */
int gc_sample(gc_handle* h, const float* cond_feats, const float* init_noise,
              const float* sigmas, int32_t n, int32_t skip_dead_call,
              float* out, gc_sample_stats* stats);

/*
 * Device-resident variants for callers that keep state in HBM between calls
 * (ensemble / autoregressive drivers, bench.py).
 *   gc_upload_cond      copy cond_feats [G,B,c_in] into the handle (H2D)
 *   gc_upload_cond_dev  same from a DEVICE pointer on this GPU (e.g. the buffer an
 *                       RCCL broadcast just filled), D2D on the handle's stream
 *   gc_sample_resident  run the sampler on the resident cond_feats and the noise set with
 *                       gc_upload_noise; asynchronous unless `stats` is non-NULL (then it waits
 *                       for the loop to read its HIP-event time)
 *   gc_upload_noise     copy init_noise [G,B,c_out] into the handle (H2D)
 *   Host buffers handed to gc_upload_cond / gc_upload_noise / gc_rollout_advance are copied into
 *   pinned staging memory owned by the handle before the call returns: the caller may reuse them
 *   at once.
 *   gc_download_sample  copy the last sample back (D2H), synchronising the stream
 *   gc_sync             wait for the handle's stream
 * Ordering while a resident sample's f16x3 domain check is still unresolved (gc_sample_resident without stats has
 * returned, gc_download_sample / gc_sync / gc_rollout_advance not yet called): the possible exact-f32 re-run of that
 * sample must see the inputs it was drawn from.  gc_upload_noise / gc_noise_draw for the NEXT sample are free -- the
 * initial noise is double-buffered -- so "sample, upload the next member's noise, download" pipelines without a
 * wait.  Every other entry point that changes sampler inputs (gc_upload_cond, gc_upload_cond_dev, gc_commit_cond,
 * gc_comm_broadcast_cond, gc_set_noisy_slots, gc_set_churn, gc_noise_seed) first resolves the pending check, i.e.
 * waits for the stream.  A second gc_sample_resident before any of these discards the first sample's check with
 * its result.
 */
int gc_upload_cond(gc_handle* h, const float* cond_feats);
int gc_upload_cond_dev(gc_handle* h, const void* cond_feats_dev);
int gc_upload_noise(gc_handle* h, const float* init_noise);
int gc_sample_resident(gc_handle* h, const float* sigmas, int32_t n, int32_t skip_dead_call,
                       gc_sample_stats* stats);
int gc_download_sample(gc_handle* h, float* out);
int gc_sync(gc_handle* h);
/*
 * Download overlapped with the next step (autoregressive drivers): gc_stash_sample waits for the last sample,
 * resolves its domain check and snapshots it into a second device buffer (a stream-ordered device-to-device copy);
 * the handle can then be given the context update and the next sample at once, and gc_download_stash copies the
 * snapshot to the host on a side stream WHILE they run (returns when `out` is complete).  Replaces nothing in the
 * reference (`jax.device_get` there, common/rollout.py:357-360); it removes the host copy from the gap between two
 * forecast steps.
 */
int gc_stash_sample(gc_handle* h);
int gc_download_stash(gc_handle* h, float* out);

/*
 * Device pointer of the resident cond_feats buffer ([G,B,c_in] float32), so a
 * collective library can write into it directly (rank != 0 receives the RCCL
 * broadcast there).  Call gc_commit_cond afterwards to re-pack it.
 */
int gc_cond_device_ptr(gc_handle* h, void** ptr, int64_t* nbytes);
int gc_commit_cond(gc_handle* h);

/*
 * Spherical white noise on the device and stochastic churn.
 * Replaces: spherical_white_noise_like / sample (gencast/samplers_utils.py:250-346) for the initial
 * state (dpm_solver_plus_plus_2s.py:71-78) and apply_stochastic_churn (samplers_utils.py:434-452) inside
 * the solver loop (dpm_solver_plus_plus_2s.py:128-137; the reference's array version of that call is
 * missing, the Dataset version defines the arithmetic).
 *   gc_noise_set_tables  static tables of the inverse real-spherical-harmonic transform on the model's
 *                        lat/lon grid (n_lat * n_lon = G; node = lat_i * n_lon + lon_j):
 *                          legendre  [lmax][n_lat][lmax]  Pn[m][lat][l] = normalised P_l^m(sin lat) *
 *                                    sqrt(4 pi p_l / (2l+1)), zero for l < m
 *                          cos_table [n_lon][lmax], sin_table [n_lon][lmax]   (sqrt(2) folded in for m > 0)
 *                        (gencast-flax-nnx_amd/noise.py builds them)
 *   gc_noise_seed        Philox4x32-10 key and the stream the next field uses; every field drawn
 *                        (initial noise or churn) advances the stream by one
 *   gc_noise_draw        fills the handle's initial-noise buffer with a fresh unit-variance field
 *                        [G, B, c_out] (instead of gc_upload_noise)
 *   gc_download_noise    copies the initial-noise buffer back (tests)
 *   gc_set_churn         per-step churn rates (stochastic_churn_rate_schedule, samplers_utils.py:415-431)
 *                        and noise_level_inflation_factor for the following gc_sample* calls; n must
 *                        equal their number of steps; n = 0 or all-zero rates switch churn off.
 *                        Step i with rate > 0:  s' = s_i (1 + rate),
 *                        x += noise * sqrt(max(s'^2 - s_i^2, 0)) * inflation, then the 2S step runs from s'.
 */
int gc_noise_set_tables(gc_handle* h, int32_t n_lat, int32_t n_lon, int32_t lmax, const float* legendre,
                        const float* cos_table, const float* sin_table);
int gc_noise_seed(gc_handle* h, uint64_t seed, uint64_t stream);
int gc_noise_draw(gc_handle* h);
int gc_download_noise(gc_handle* h, float* out);
int gc_set_churn(gc_handle* h, const float* rates, int32_t n, float noise_level_inflation_factor);

/*
 * Ensemble exchange.  Replaces: the replication of inputs / forcings over the local
 * devices in chunked_prediction_generator_multiple_runs (common/rollout.py:41-75 `_replicate_dataset`,
 * :123-139 `device_put_sharded`); members then run independently, one per GPU (:312-322), and are
 * pulled back per device (:357-360 -> gc_download_sample on every rank).
 * One process (or thread) per GPU, each with its own handle.  The ONLY collective on the path is
 * the broadcast of the packed conditioning [G, B, c_in] from `root`, issued by the library itself:
 * ncclBroadcast (RCCL, over xGMI between the GPUs of a node) in place on the handle's resident
 * buffer and on the handle's stream, followed by the re-pack -- no host copy, no torch.  Nothing
 * inside the denoiser or the sampler communicates.
 *   gc_comm_unique_id      rank 0: a GC_COMM_ID_BYTES blob (ncclUniqueId); hand it to every rank by
 *                          any means (file, socket, environment)
 *   gc_comm_init           collective over all ranks: ncclCommInitRank on this handle's device
 *   gc_comm_broadcast_cond collective: root's resident conditioning (gc_upload_cond) -> every rank
 *   gc_comm_allreduce_max  collective: *value <- max over ranks (benchmark timing); also a barrier
 *   gc_comm_info           what RCCL itself says about this handle's communicator: ncclCommCount / ncclCommUserRank
 *                          (0 ranks, rank -1 without a communicator) -- lets a driver verify that the exchange it
 *                          reports really spans all its ranks
 * librccl.so.1 is loaded at the first gc_comm_* call (GC_RCCL_LIBRARY overrides the path); failures
 * return GC_ERR_COMM.
 */
#define GC_COMM_ID_BYTES 128
int gc_comm_unique_id(void* id_out /* [GC_COMM_ID_BYTES] */);
int gc_comm_init(gc_handle* h, const void* id, int32_t rank, int32_t world_size);
int gc_comm_info(gc_handle* h, int32_t* num_ranks, int32_t* rank);
int gc_comm_broadcast_cond(gc_handle* h, int32_t root);
int gc_comm_allreduce_max(gc_handle* h, double* value);
int gc_comm_destroy(gc_handle* h);

/*
 * Autoregressive context update on the device.
 * Replaces, for the packed conditioning: the context roll of autoregressive_rollout
 * (training/train_helpers.py:596-622: drop the oldest frame, append the predicted one, carry
 * input-only variables, take forcing variables from this step's forcings) together with
 * InputsAndResiduals' un-normalise + add-last-input and the re-normalisation of the next step
 * (common/normalization.py:100-121,200-238), which in normalised space is one affine per channel.
 * For every conditioning channel c of [G, B, c_in] (plan arrays have c_in entries):
 *   kind 0  keep       new[c] = old[c]
 *   kind 1  copy       new[c] = old[src[c]]
 *   kind 2  residual   new[c] = old[src[c]] + a[c] * sample[sidx[c]] + b[c]
 *   kind 3  forcing    new[c] = forcings[sidx[c]]          (row-wise, forcings is [G, B, n_forcing])
 *   kind 4  direct     new[c] = a[c] * sample[sidx[c]] + b[c]
 * `sample` is the last sample held by the handle (gc_sample_resident / gc_sample).
 * gc_rollout_advance applies the plan to all rows in one launch and re-packs the conditioning;
 * `forcings` is a HOST array (NULL allowed when the plan has no kind 3).
 */
int gc_rollout_plan(gc_handle* h, const int32_t* kind, const int32_t* src, const int32_t* sidx,
                    const float* a, const float* b, int32_t n_forcing);
int gc_rollout_advance(gc_handle* h, const float* forcings);
/* Copies the resident conditioning [G, B, c_in] back to the host (tests, checkpoints). */
int gc_download_cond(gc_handle* h, float* out);

/*
 * Measurement support (bench.py, rocprof cross-check).  Kernel classes are
 * indexed 0..gc_num_kernel_classes()-1; gc_kernel_class_name gives the label
 * that also prefixes the HIP kernel symbol.  With profiling enabled on a class,
 * every launch of that class is bracketed by HIP events on the handle's stream;
 * gc_profile_read synchronises and returns launches and total milliseconds.
 */
int         gc_num_kernel_classes(void);
const char* gc_kernel_class_name(int cls);
int gc_profile_enable(gc_handle* h, int cls /* -1 = off */);
/* Bracket only every `stride`-th launch of the profiled class (default 1) so that the event
 * records do not perturb a timed region. */
int gc_profile_set_stride(gc_handle* h, int stride);
int gc_profile_read(gc_handle* h, int32_t* launches, float* total_ms);
/* Algorithmic FLOPs and compulsory HBM bytes of ONE denoiser call for this
 * handle's configuration (formulas in DESIGN.md). */
int gc_algorithmic_work(gc_handle* h, double* flops, double* bytes);
/* Named counters: "range_fallbacks" (calls re-run on the f32 kernels by the f16x3 domain guard),
 * "launches_per_call" (kernel launches of the last denoiser forward), "weights_f16_unsafe",
 * "graph_captures" / "graph_replays" (sampler graphs captured / samples launched as one hipGraphLaunch),
 * "fp16_storage" (1 when the last forward kept its activations as 2-byte fp16 arrays in HBM: features = f16 on the
 * f16x3 weight-streaming kernels; 0 when it ran on float32 containers), "split_edge" (1 when the edge MLPs run with their first layer split by input block: on
 * from latent 512), "attention_items" (work items of the last call's attention launches when they ran as a host-made item list -- one whole
 * query tile per CU, then the remaining tiles as key-range pieces merged by the out-projection: the 1-degree size, 512; 0: plain launch),
 * "m2g_fused_sum" (1 when the last forward added every grid node's three updated mesh2grid edges inside the edge MLP's epilogue --
 * jraph.segment_sum of common/typed_graph_net.py:175-182 without storing the edges; 0: edge update + a segment-sum launch, the form
 * any mesh2grid edge set with other in-degrees than 3 takes), "embed_cache" (samples so far whose grid embedding ran on the cached
 * per-sample-constant part of its first layer: only the c_out noisy-target columns are multiplied per call, the other 3 + c_in - c_out
 * once per sample -- dpm_solver_plus_plus_2s.py:107-112 closes over them; float32 node features, hidden_layers = 1). */
int gc_get_counter(gc_handle* h, const char* name, int64_t* value);

#ifdef __cplusplus
}
#endif
#endif /* GENCAST_HIP_H_ */
