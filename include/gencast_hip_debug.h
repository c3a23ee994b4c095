/*
 * gencast_hip_debug.h -- test-only entry points of libgencast_hip.so.
 *
 * Not part of the drop-in boundary: these exist so the parity tests can compare
 * every stage of the denoiser against the oracle's intermediates
 * (oracle/gencast_oracle.py: denoiser_forward(return_intermediates=True)).
 */
#ifndef GENCAST_HIP_DEBUG_H_
#define GENCAST_HIP_DEBUG_H_

#include <stdint.h>

#include "gencast_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Copies an intermediate buffer of the most recent forward to the host, in the
 * CALLER's mesh numbering.  Names: cond, g0, m0, e1, agg1, x (transformer state
 * after the layers that ran), h (the last LayerNorm + conditioning output inside the blocks), g1, qkv, att, m2,
 * f1, agg2, g2, y, the static
 * embeddings m0_hat, e0_hat, f0_hat, and "cond:<parameter path of a conditional_linear_layer>" = the
 * [B, 2n] vectors [scale (the +1 included) | offset] that conditioning linear produced in the last gc_denoise.  With out == NULL only rows/cols are
 * returned.
 */
int gc_debug_fetch(gc_handle* h, const char* name, float* out, int64_t capacity, int64_t* rows,
                   int64_t* cols);

/* Run only the first `num_layers` transformer blocks in later forwards (-1 = all). */
int gc_debug_set_layer_limit(gc_handle* h, int32_t num_layers);

/*
 * Later forwards RETURN inside transformer block `layer` (0-based; -1 = off), after phase
 *   0  the pre-attention row pass       (x = residual stream entering the block, h = cond(LN(x)))
 *   1  the QKV projection               (qkv)
 *   2  attention + out-projection + row pass  (x = stream after the attention half, h = cond(LN(x)))
 * so that gc_debug_fetch sees what that phase wrote ("h" is otherwise overwritten inside the block).  The network
 * output of such a forward is whatever an earlier forward left: run one complete forward first.
 */
int gc_debug_set_stop(gc_handle* h, int32_t layer, int32_t phase);

/* Internal mesh order: perm_out[new_id] = caller_id ([M]). */
int gc_debug_mesh_permutation(gc_handle* h, int32_t* perm_out);

/* Attention tiling statistics: tiles, 32-key chunks over all tiles, k-hop nnz. */
int gc_debug_attention_stats(gc_handle* h, int64_t* n_tiles, int64_t* n_chunks, int64_t* khop_nnz);

#ifdef __cplusplus
}
#endif
#endif /* GENCAST_HIP_DEBUG_H_ */
