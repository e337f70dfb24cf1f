/*
 * glnn_hip.h -- C ABI of libglnn_hip.so, the MI355X (gfx950) implementation of the GLNN hot path:
 * teacher neighbour aggregation + dense projections, and the MLP student distillation step.
 *
 * The reference (snap-research/graphless-neural-networks) has NO native/FFI layer: it is pure
 * Python whose graph arithmetic is delegated to dgl==0.6.1 and whose dense arithmetic is
 * torch==1.7.0.  Each entry point below therefore cites the reference *call site* (file:line under
 * /root/reference) whose arithmetic it replaces; INTEGRATION.md shows the ctypes binding a
 * maintainer would add at that call site.
 *
 * Conventions (all entry points):
 *   - plain pointers + explicit sizes and leading dimensions (in ELEMENTS), no torch types;
 *   - every data pointer is a DEVICE pointer (HBM) unless the parameter says "host";
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls only ENQUEUE
 *     work, never synchronise, never allocate or free caller memory;
 *   - return value: 0 = GLNN_OK, negative = glnn_status; glnn_last_error() gives a thread-local
 *     message for the last failing call on this thread;
 *   - fp32 values, int32 column indices, int64 row pointers; row-major matrices;
 *   - feature matrices must have a leading dimension that is a multiple of 4 floats and a
 *     16-byte aligned base (the kernels move float4); columns [d, ld) are padding: read but
 *     never trusted, and written as 0 by kernels that own an output.
 */
#ifndef GLNN_HIP_H
#define GLNN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLNN_API __attribute__((visibility("default")))

typedef enum {
  GLNN_OK = 0,
  GLNN_ERR_INVALID_ARG = -1,   /* null pointer, bad size, bad alignment / leading dimension */
  GLNN_ERR_UNSUPPORTED = -2,   /* shape or mode outside what the kernels implement */
  GLNN_ERR_HIP = -3,           /* a HIP runtime call or launch failed */
  GLNN_ERR_NO_DEVICE = -4      /* no gfx950 device visible */
} glnn_status;

GLNN_API int glnn_abi_version(void);               /* bumped on any signature change */
GLNN_API const char* glnn_last_error(void);        /* thread-local, never NULL */
GLNN_API int glnn_device_info(int* cu_count, int* xcd_count, char* arch_buf, int arch_buf_len);
/* sizeof of the descriptor structs below as THIS build sees them (0 = glnn_mlp_step_desc, 1 = glnn_sage_step_desc,
 * 2 = glnn_sage_layer, 3 = glnn_adam_desc; -1 otherwise): lets a binding in another language check its mirror of the layout at load time. */
GLNN_API int64_t glnn_struct_bytes(int which);
/* The library reads its GLNN_* environment switches (csrc/glnn_common.h, glnn::Options: choices between supported, equal-result
 * forms of a launch sequence, for tests and A/B timing) ONCE, at the first call.  This re-reads them; not for concurrent use. */
GLNN_API void glnn_reload_options(void);

/* ------------------------------------------------------------------------------------------
 * K1/K2  CSR neighbour aggregation (SpMM with an implicit all-ones adjacency, multi-edges kept).
 * CSR is over DESTINATION rows: row v lists the sources u of its in-edges u->v.
 *
 * mode GLNN_AGG_SUM       out[v,:] = row_scale[v] * sum_{u->v} col_scale[u] * x[u,:]
 *        replaces g.update_all(fn.copy_u("h","m"), fn.sum("m","h")), reference utils.py:185
 *        (feature_prop) and the aggregation inside dgl GraphConv(norm="both"), reference
 *        models.py:193 (ctor :170-187): col_scale = out_deg.clamp(1)^-1/2,
 *        row_scale = in_deg.clamp(1)^-1/2.  Either scale may be NULL (= 1).
 * mode GLNN_AGG_SAGE_GCN  out[v,:] = (sum_{u->v} x[u,:] + x_self[v,:]) / (in_deg(v) + 1)
 *        replaces the "gcn" aggregator of dgl SAGEConv, reference models.py:112,138
 *        (ctor :84-99); x_self is h_dst = the first n_dst rows of the block's source features
 *        (models.py:109,137), normally x_self == x.  self_rows (optional, SAGE_GCN only): the self row of
 *        destination v is x_self[self_rows[v]] -- with `indices` holding GLOBAL node ids and x = x_self = the whole
 *        feature matrix this folds `batch_feats = feats[input_nodes]` (train_and_eval.py:42) into the first layer's
 *        gather: the outermost block of a training batch is aggregated straight out of `feats`.
 * Epilogue (both modes), per output element, in this order, each optional:
 *        y = y * ep_scale[j] + ep_shift[j] ;  y = max(y, 0) if relu
 *   (bias, eval-mode BatchNorm and ReLU of models.py:139-143 when the dense projection was
 *    applied BEFORE aggregation; NULL/0 otherwise).
 * n_src bounds the column indices (only used for argument checking).  Rows wider than 256 floats
 * are processed in column tiles of 256 (one launch per tile).
 * ------------------------------------------------------------------------------------------ */
#define GLNN_AGG_SUM 0
#define GLNN_AGG_SAGE_GCN 1

GLNN_API int glnn_spmm_csr_f32(const int64_t* indptr, const int32_t* indices, int64_t n_dst,
                               int64_t n_src, const float* x, int64_t ldx, int d, int mode,
                               const float* row_scale, const float* col_scale,
                               const float* x_self, int64_t ld_self, const int64_t* self_rows,
                               const float* ep_scale, const float* ep_shift, int relu, float* out,
                               int64_t ldo, void* stream);

/* K1F  Fused SAGE-"gcn" layer for aggregate-first layers (d_in, d_out <= 256):
 *   out[v,:] = epi( ((sum_{u->v} x[u,:] + x_self[v,:]) / (in_deg(v)+1)) @ W^T ),  epi = *ep_scale +ep_shift, ReLU
 * = SAGEConv(in,out,"gcn") + bias + eval BatchNorm + ReLU of reference models.py:138-143 in ONE launch; the
 * aggregated rows stay in LDS and feed the fp32 MFMA directly.  W must be pre-packed into MFMA fragment
 * order with glnn_pack_weight_f32 (glnn_packed_weight_floats(d_out, d_in) floats, 16-byte aligned). */
GLNN_API int64_t glnn_packed_weight_floats(int d_out, int d_in);
GLNN_API int glnn_pack_weight_f32(const float* w, int64_t ldw, int d_out, int d_in, float* w_packed,
                                  void* stream);
/* Optional chained projection (w2_packed != NULL: W2 [d_out2, d_out] packed like W): additionally
 *   out2[v,:] = out[v,:] @ W2^T   (no epilogue)
 * i.e. the dense half of the NEXT layer when that layer projects first (in > out, models.py:138 on the last layer of the
 * products / arxiv teachers): the hidden row goes from the MFMA accumulators through LDS into a second MFMA pass; with
 * out == NULL it never reaches HBM at all. */
GLNN_API int glnn_sage_fused_f32(const int64_t* indptr, const int32_t* indices, int64_t n_dst,
                                 int64_t n_src, const float* x, int64_t ldx, int d_in,
                                 const float* x_self, int64_t ld_self, const float* w_packed,
                                 int d_out, const float* ep_scale, const float* ep_shift, int relu,
                                 float* out, int64_t ldo, const float* w2_packed, int d_out2,
                                 float* out2, int64_t ldo2, const int32_t* tile_order, void* stream);
/* tile_order (optional, ceil(n_dst / 32) entries: a permutation of the 32-row tile ids): workgroup i takes tile tile_order[i].  A caller
 * that sorts the tiles by their heaviest row (descending in-degree) starts the hub rows of a power-law graph first instead of wherever
 * their ids put them -- what matters for SHORT launches (a row shard's chunk: one 8 k-edge hub row is 100 us of a 0.7 ms launch). */

/* ABI 9: HUB ROWS split over workgroups.  A destination row of more than glnn_hub_row_threshold() in-edges is always summed in segments
 * of glnn_hub_segment_edges() edges, 1/8 of a segment per wave: a wave's share of the row is the sum over the segments (ascending) of its
 * piece of each, gathered into a fresh accumulator; the eight shares are folded like the wave partials of any long row.  The pieces are
 * gathered by the one workgroup that owns the row, or, given a plan, by one workgroup PER SEGMENT in a launch of its own in front of the
 * aggregation, whose partial sums the row's owner reads back.  Same bits either way; what changes is the tail of SHORT launches: a
 * 17 k-edge row of the products graph keeps one workgroup busy for ~0.5 ms at d = 256, most of a row shard's 0.7 ms chunk launch
 * (scripts/hub_tail_probe.py).  The plan is per (indptr, n_dst) launch range and caller-built (one pass over the degrees:
 * glnn_amd.ops.hub_plan):
 *   rows      ascending ids v (relative to `indptr`) of the rows with indptr[v+1] - indptr[v] > threshold; a hub row that is not listed
 *             is summed by its owner; a listed row must have exactly ceil(degree / segment_edges) segments
 *   seg_ptr   [n_hub + 1] running segment count (seg_ptr[0] = 0, seg_ptr[n_hub] = n_seg)
 *   slab      [slab_rows >= 8 n_seg][ld_slab >= d rounded up to 4] floats of scratch (row 8 s + w = wave w's piece of segment s), 16-byte
 *             aligned; overwritten by every call that is given the plan (one stream at a time)
 * Column-tiled launches (d > 256) ignore the plan.  The reference has no counterpart (dgl's SpMM is one launch, models.py:112,138). */
typedef struct glnn_hub_plan {
  const int64_t* rows;
  const int32_t* seg_ptr;
  int32_t n_hub, n_seg;
  float* slab;
  int64_t ld_slab, slab_rows;
} glnn_hub_plan;
GLNN_API int glnn_hub_row_threshold(void);
GLNN_API int glnn_hub_segment_edges(void);
GLNN_API int glnn_spmm_csr_plan_f32(const int64_t* indptr, const int32_t* indices, int64_t n_dst,
                                    int64_t n_src, const float* x, int64_t ldx, int d, int mode,
                                    const float* row_scale, const float* col_scale,
                                    const float* x_self, int64_t ld_self, const int64_t* self_rows,
                                    const float* ep_scale, const float* ep_shift, int relu, float* out,
                                    int64_t ldo, const glnn_hub_plan* plan, void* stream);
GLNN_API int glnn_sage_fused_plan_f32(const int64_t* indptr, const int32_t* indices, int64_t n_dst,
                                      int64_t n_src, const float* x, int64_t ldx, int d_in,
                                      const float* x_self, int64_t ld_self, const float* w_packed,
                                      int d_out, const float* ep_scale, const float* ep_shift, int relu,
                                      float* out, int64_t ldo, const float* w2_packed, int d_out2,
                                      float* out2, int64_t ldo2, const int32_t* tile_order,
                                      const glnn_hub_plan* plan, void* stream);

/* ABI 12 (round 6): ONE fused launch over the CHUNKS of a row shard, with a completion signal per chunk.  A rank of the sharded forward
 * (no reference counterpart: the reference is single-device, train_teacher.py:162-165) produces its rows chunk by chunk so that each
 * chunk's rows can leave over the links while the next is computed; launched one by one the chunks pay a short launch's ramp and tail
 * each (two launches of a rank's layer 2 at N = 8: 2.55 ms, one launch over the same rows: see DESIGN.md section 6).  Here the chunks
 * are tile ranges of ONE launch:
 *   row_start  [n_chunks + 1] first row (relative to `indptr`) of every chunk, multiples of 32 (row_start[0] = 0); row_start[n_chunks]
 *              >= n_dst (a chunk may be short or empty: an EMPTY chunk is never signalled)
 *   self_row   row of `x_self` holding the self row of row_start[c] (the chunk's self rows are consecutive from there)
 *   out_row    row of `out` / `out2` that receives row_start[c]
 *   arrivals   n_chunks device counters, zero before the first launch (the launch leaves them zero)
 *   signal     n_chunks words from glnn_signal_alloc(); when every row of chunk c is stored (acknowledged by the L2) the kernel
 *              stores `epoch` there -- glnn_stream_wait_value32(other_stream, signal[c], epoch) then holds that stream's next
 *              operation (the chunk's all-gather) until the chunk is complete AND written back to device memory, while the launch
 *              is still running.  Use a new, larger epoch per launch (the wait is "value >= epoch", unsigned 32-bit).
 * `tile_order` may permute the tiles freely; completion order follows it (the caller concatenates the chunks' orders). */
#define GLNN_MAX_CHUNKS 8
typedef struct glnn_chunk_signals {
  int32_t n_chunks;
  int64_t row_start[GLNN_MAX_CHUNKS + 1];
  int64_t self_row[GLNN_MAX_CHUNKS];
  int64_t out_row[GLNN_MAX_CHUNKS];
  int32_t* arrivals;
  uint32_t* signal[GLNN_MAX_CHUNKS];
  uint32_t epoch;
} glnn_chunk_signals;
GLNN_API int glnn_sage_fused_chunks_f32(const int64_t* indptr, const int32_t* indices, int64_t n_dst,
                                        int64_t n_src, const float* x, int64_t ldx, int d_in,
                                        const float* x_self, int64_t ld_self, const float* w_packed,
                                        int d_out, const float* ep_scale, const float* ep_shift, int relu,
                                        float* out, int64_t ldo, const float* w2_packed, int d_out2,
                                        float* out2, int64_t ldo2, const int32_t* tile_order,
                                        const glnn_hub_plan* plan, const glnn_chunk_signals* chunks, void* stream);
/* The stand-alone SAGE-"gcn" aggregation in the same form (glnn_spmm_csr_plan_f32 + chunks; mode = GLNN_AGG_SAGE_GCN, d <= 256): the self
 * row of row v of chunk c is x_self[self_rows ? self_rows[v] : self_row[c] + v - row_start[c]], its output row out_row[c] + v - row_start[c].
 * Rows of more than the long-row threshold are summed by the launch's long-row workgroups chunk by chunk; a chunk signals when its row
 * workgroups AND every long-row workgroup's share of it are stored. */
GLNN_API int glnn_spmm_csr_chunks_f32(const int64_t* indptr, const int32_t* indices, int64_t n_dst,
                                      int64_t n_src, const float* x, int64_t ldx, int d, int mode,
                                      const float* row_scale, const float* col_scale,
                                      const float* x_self, int64_t ld_self, const int64_t* self_rows,
                                      const float* ep_scale, const float* ep_shift, int relu, float* out,
                                      int64_t ldo, const glnn_hub_plan* plan, const glnn_chunk_signals* chunks,
                                      void* stream);
/* A 32-bit signal word a stream can wait on (hipExtMallocWithFlags(hipMallocSignalMemory)), initialised to 0; its current value;
 * "the next operation of `stream` starts when *signal >= value" (hipStreamWaitValue32, GTE) -- followed by an empty kernel on `stream`,
 * whose end-of-kernel release writes back the L2s in which the signalled rows may still sit. */
GLNN_API int glnn_signal_alloc(uint32_t** signal);
GLNN_API int glnn_signal_free(uint32_t* signal);
GLNN_API int glnn_signal_read(const uint32_t* signal, uint32_t* value);
GLNN_API int glnn_stream_wait_value32(void* stream, uint32_t* signal, uint32_t value);

/* in_deg[v] = t(indptr[v+1]-indptr[v]); out_deg[u] = t(#edges with source u), as floats, t = `transform`:
 *   GLNN_DEG_RAW          the degree itself            g.in_degrees() / g.out_degrees()
 *   GLNN_DEG_RSQRT_CLAMP1 deg.clamp(min=1) ** -0.5     the norm of dgl GraphConv(norm="both") and utils.py:178-179
 *   GLNN_DEG_INV_PLUS1    1 / (deg + 1)                the divisor of the SAGE-"gcn" aggregator (its backward's col_scale)
 * Either output may be NULL.  nnz = indptr[n_dst] (host value).  out_deg need not be zeroed by the caller. */
#define GLNN_DEG_RAW 0
#define GLNN_DEG_RSQRT_CLAMP1 1
#define GLNN_DEG_INV_PLUS1 2
GLNN_API int glnn_degrees_f32(const int64_t* indptr, const int32_t* indices, int64_t n_dst,
                              int64_t n_src, int64_t nnz, int transform, float* in_deg,
                              float* out_deg, void* stream);

/* ------------------------------------------------------------------------------------------
 * K3  Dense projection on the fp32 MFMA path (v_mfma_f32_32x32x2_f32, exact fp32).
 *   C[m,n] = epilogue( sum_k A'[m,k] * B[k,n] )
 *   A' = A, or rows gathered through a_rows (A'[m,:] = A[a_rows[m],:], replaces feats[idx]
 *        at reference train_and_eval.py:76 / models.py:136), optionally passed through the
 *        "previous layer tail"  A'[m,k] = drop(max(A[m,k]*a_scale[k] + a_shift[k], 0))  (BatchNorm,
 *        ReLU and training-mode Dropout of reference models.py:48-52 folded into the operand load;
 *        a_scale NULL = identity, no ReLU, no dropout).  drop(v) = keep(m,k) ? v/(1-drop_p) : 0 with
 *        keep() the counter-based mask of glnn_dropout_mask_u8 (drop_p = 0 disables it).
 *   B  = W^T with W [n,k] row-major (torch.nn.Linear / dgl SAGEConv.fc_neigh, b_layout 0)
 *        or W [k,n] row-major (dgl GraphConv.weight, b_layout 1).
 *   epilogue, in order, each optional: * row_scale[m] ; * ep_scale[n] ; + ep_shift[n] ; ReLU.
 *   workspace (optional, floats): lets skinny outputs (few 128x128 tiles, deep K) split the K
 *   reduction over more workgroups; partials are summed in fixed order (deterministic).  NULL = never.
 * replaces fc_neigh / GraphConv weight / nn.Linear: reference models.py:45,112,138,193.
 * ------------------------------------------------------------------------------------------ */
GLNN_API int glnn_gemm_f32(const float* a, int64_t lda, const int64_t* a_rows,
                           const float* a_scale, const float* a_shift, float drop_p,
                           uint32_t drop_seed, int64_t m, int k,
                           const float* b, int64_t ldb, int b_layout, int n,
                           const float* row_scale, const float* ep_scale, const float* ep_shift,
                           int relu, float* c, int64_t ldc, float* workspace,
                           int64_t workspace_floats, void* stream);

/* TN form, the weight-gradient shape:  C[i,j] = sum_m A[m,i] * B'[m,j]   (i < ka, j < nb)
 *   dW[out,in] = dZ^T @ A_prev with A = dZ [m,out] and B = the previous layer's PRE-activation
 *   (or the gathered input features), B' = B passed through the same operand transform as
 *   glnn_gemm_f32 (row gather b_rows, then max(B*b_scale+b_shift,0)), so the activation is
 *   recomputed instead of stored.  Replaces the nn.Linear weight/bias gradient autograd computes
 *   inside loss.backward(), reference train_and_eval.py:84.
 *   col_sum_a != NULL also writes col_sum_a[i] = sum_m A[m,i] (the bias gradient).
 *   workspace (floats) is used for a deterministic split of the m-reduction when the output is
 *   small, and for the column sums (needs >= 64*ka floats for those). */
GLNN_API int glnn_gemm_tn_f32(const float* a, int64_t lda, int64_t m, int ka, const float* b,
                              int64_t ldb, const int64_t* b_rows, const float* b_scale,
                              const float* b_shift, float drop_p, uint32_t drop_seed, int nb,
                              float* c, int64_t ldc,
                              float* col_sum_a, float* workspace, int64_t workspace_floats,
                              void* stream);

/* ------------------------------------------------------------------------------------------
 * K4  Losses on raw logits, forward + gradient wrt logits in one pass.
 *   kind GLNN_LOSS_NLL: NLLLoss()(log_softmax(z), y)                 train_student.py:278
 *   kind GLNN_LOSS_KL : KLDivLoss("batchmean", log_target=True)(log_softmax(z), t)  :279
 *   loss_out[0]   = unscaled mean loss of the batch (what .item() reports, train_and_eval.py:80)
 *   loss_accum[0]+= the same (optional, NULL to skip) so a whole pass reads back ONE scalar
 *   dlogits       = d(lamb * loss)/dz   (train_and_eval.py:82 scales the loss by lamb)
 *   logprob_out   = log_softmax(z) (optional)
 *   workspace     >= min(ceil(rows/4), 1024) floats (per-workgroup partial sums; fixed-order reduce)
 *   labels / target_logp may be indexed through label_rows / target_rows (the batch's node ids) so
 *   that labels[idx] / out_t[idx] (train_and_eval.py:79) are never materialised; NULL = identity.
 * ------------------------------------------------------------------------------------------ */
#define GLNN_LOSS_NLL 0
#define GLNN_LOSS_KL 1
GLNN_API int glnn_softmax_loss_f32(const float* logits, int64_t ldz, int64_t rows, int c, int kind,
                                   const int64_t* labels, const int64_t* label_rows,
                                   const float* target_logp, int64_t ldt, const int64_t* target_rows,
                                   float lamb, float* dlogits, int64_t ldg, float* logprob_out,
                                   int64_t ldl, float* loss_out, float* loss_accum,
                                   float* workspace, int64_t workspace_floats, void* stream);

/* The last layer of a student and its criterion in ONE launch (ABI 11; reference models.py:45-52 last Linear + train_and_eval.py:77-84):
 *   logits = tail(a) . w^T + bias,  tail(a) = dropout(relu(a * a_scale + a_shift)) (a_scale / a_shift NULL: a is used as stored),
 *   followed by glnn_softmax_loss_f32's arithmetic on those logits (same loss / dlogits bits), by workgroups that own 16 rows.
 *   For LARGE batches in front of a NARROW classifier: rows > 1024, c <= 48, k % 256 == 0, 256 <= k <= 4096, float4-addressable rows of
 *   a and w; kind < 0: logits only (dlogits / loss_out / workspace unused).  With a criterion: rows <= 4096 and
 *   workspace >= ceil(rows/4) floats.  Any other shape: GLNN_ERR_UNSUPPORTED with nothing launched (use glnn_gemm_f32 +
 *   glnn_softmax_loss_f32). */
GLNN_API int glnn_classifier_loss_f32(const float* a, int64_t lda, const float* a_scale, const float* a_shift,
                                      float drop_p, uint32_t drop_seed, int64_t rows, int k, const float* w, int64_t ldw,
                                      int c, const float* bias, float* logits, int64_t ldz, int kind,
                                      const int64_t* labels, const int64_t* label_rows, const float* target_logp,
                                      int64_t ldt, const int64_t* target_rows, float lamb, float* dlogits, int64_t ldg,
                                      float* loss_out, float* loss_accum, float* workspace, int64_t workspace_floats,
                                      void* stream);

/* row-wise log_softmax only (evaluate paths, train_and_eval.py:98,124). in place allowed. */
GLNN_API int glnn_log_softmax_f32(const float* logits, int64_t ldz, int64_t rows, int c, float* out,
                                  int64_t ldo, void* stream);

/* ------------------------------------------------------------------------------------------
 * K5  BatchNorm1d training statistics + backward pieces (reference models.py:28-31,48-49:
 *     eps 1e-5, momentum 0.1, biased batch variance for normalisation, unbiased for running_var).
 *   glnn_bn_stats_f32: from z [rows,h] computes a_scale = gamma*rstd, a_shift = beta - mean*a_scale
 *     (the operand transform the next glnn_gemm_f32 consumes), saves mean/rstd, and updates
 *     running_mean / running_var / num_batches_tracked in place.  workspace: c = ceil(rows/128) row chunks ->
 *     >= 2*c*h floats, plus 3*ceil(c/64)*h when c > 256 (two-level combine of the per-chunk statistics).
 *   glnn_bn_relu_bwd_f32: given da (grad wrt the post-ReLU activation) and z, computes
 *     dy = da * [z*a_scale+a_shift > 0], dgamma = sum dy*xhat, dbeta = sum dy and
 *     dz = gamma*rstd*(dy - mean(dy) - xhat*mean(dy*xhat)); in place on da allowed.
 *     With gamma == NULL (norm_type "none") it is the plain ReLU backward dz = da*[z>0].
 *     dz_col_sum (optional) receives sum over rows of dz = the bias gradient of the Linear in front.
 *     workspace: >= ceil(rows/128)*h*(2 if gamma else 0 + 1 if dz_col_sum else 0) + 2*h floats.
 *     drop_p > 0 first applies the Dropout backward da *= keep(row,col)/(1-drop_p) with the same
 *     (drop_seed) mask the forward operand transform used.
 * ------------------------------------------------------------------------------------------ */
GLNN_API int glnn_bn_stats_f32(const float* z, int64_t ldz, int64_t rows, int h, const float* gamma,
                               const float* beta, float eps, float momentum, float* running_mean,
                               float* running_var, int64_t* num_batches_tracked, float* mean_out,
                               float* rstd_out, float* a_scale_out, float* a_shift_out,
                               float* workspace, int64_t workspace_floats, void* stream);

/* ABI 8: z = a W^T + bias [m, n] followed by glnn_bn_stats_f32 on z -- Linear + BatchNorm1d statistics of a hidden layer (reference
 * models.py:43-47, 110-114).  When the product runs on the row-panel kernel (k <= 128 over many rows) or the pipelined kernel
 * (k % 32 == 0, plain operands, >= 64 output tiles) the statistics' first pass comes out of the product's epilogue (per-workgroup /
 * per-tile count, mean, M2 of the stored values) and z is not read again; otherwise the two calls run as they are.  Plain operands
 * only.  ws_bn: as glnn_bn_stats_f32, and >= 3 * 128 * n floats for the row-panel form.  GLNN_GEMM_STATS=0 forces the two-call form. */
GLNN_API int glnn_linear_bn_stats_f32(const float* a, int64_t lda, int64_t m, int k, const float* w, int64_t ldw, int n,
                                      const float* bias, float* z, int64_t ldz, const float* gamma, const float* beta,
                                      float eps, float momentum, float* running_mean, float* running_var,
                                      int64_t* num_batches_tracked, float* mean_out, float* rstd_out, float* a_scale_out,
                                      float* a_shift_out, float* ws_gemm, int64_t ws_gemm_floats, float* ws_bn,
                                      int64_t ws_bn_floats, void* stream);

GLNN_API int glnn_bn_relu_bwd_f32(const float* da, int64_t ldda, const float* z, int64_t ldz,
                                  int64_t rows, int h, const float* gamma, const float* mean,
                                  const float* rstd, const float* a_scale, const float* a_shift,
                                  float drop_p, uint32_t drop_seed,
                                  float* dz, int64_t lddz, float* dgamma, float* dbeta,
                                  float* dz_col_sum,
                                  float* workspace, int64_t workspace_floats, void* stream);

/* out[j] = sum over rows of x[:, j] (fixed summation order): the bias gradient of a layer whose dz does not come out of
 * glnn_bn_relu_bwd_f32 (last GraphConv of the full-graph GCN step, reference train_and_eval.py:12-29).
 * workspace >= ceil(rows/128) * h floats. */
GLNN_API int glnn_col_sum_f32(const float* x, int64_t ldx, int64_t rows, int h, float* out,
                              float* workspace, int64_t workspace_floats, void* stream);

/* ------------------------------------------------------------------------------------------
 * K6  Fused multi-tensor Adam, torch.optim.Adam semantics (L2 weight decay folded into the
 *     gradient, bias correction, eps outside the sqrt): reference train_student.py:275-277,
 *     train_teacher.py:234-236, stepped at train_and_eval.py:85.
 *   The tensor table is 4 device arrays of `num_tensors` pointers + one of sizes (elements).
 *   `step` is the 1-based step count AFTER this update.
 * ------------------------------------------------------------------------------------------ */
GLNN_API int glnn_adam_step_f32(float* const* params, const float* const* grads,
                                float* const* exp_avg, float* const* exp_avg_sq,
                                const int64_t* sizes, int num_tensors, int64_t max_size,
                                float lr, float beta1, float beta2, float eps, float weight_decay,
                                int64_t step, void* stream);

/* ------------------------------------------------------------------------------------------
 * The whole forward + loss + backward of one student step in ONE call (reference train_and_eval.py:74-84:
 * model(None, feats[idx]) -> log_softmax -> criterion -> (loss*lamb).backward()), issuing the K3-K5
 * launches above in a fixed order.  Every buffer is caller-owned; the descriptor only carries pointers:
 *   w/b/gw/gb[l]      Linear l weight [dims[l+1], dims[l]] / bias and their gradients (contiguous)
 *   gamma..nbt[l]     BatchNorm1d after hidden layer l (l < L-1) when batchnorm != 0
 *   mean/rstd/a_scale/a_shift[l]  per-hidden-layer vectors; with batchnorm == 0 a_scale/a_shift must be
 *                     pre-filled with ones/zeros (the operand transform is then the plain ReLU)
 *   z[l] (ldz[l])     pre-activation output of hidden layer l, [max_batch, dims[l+1]]
 *   da/dz             scratch [max_batch, max hidden]; ws_*: workspaces of the individual kernels
 *   loss_out/accum    as in glnn_softmax_loss_f32
 * idx: the batch's row ids into feats (and into labels / target_logp through target_rows), NULL = rows 0..m-1.
 * drop_seeds: host array of L-1 uint32 (one per hidden layer) when dropout_p > 0.
 * The optimiser step is NOT included: call glnn_adam_step_f32 next (a gradient all-reduce may sit between).
 * ------------------------------------------------------------------------------------------ */
#define GLNN_MLP_MAX_LAYERS 8
#define GLNN_MLP_COUNTERS 1024   /* ints behind glnn_mlp_step_desc.sync_counters: one per 64-column block + one for the loss */

/* Optional data-parallel hook (SURVEY.md 8e "Student"): when one batch is split over `world` ranks the
 * BatchNorm batch statistics must be taken over the WHOLE batch for the step to equal the single-GPU
 * reference step.  The library leaves the collective to the host: it fills `send` with `floats` values,
 * calls exchange(ctx, send, recv, floats, stream) and expects recv = the concatenation of every rank's
 * `send` in rank order (an all-gather, enqueued on / ordered after `stream`; ncclAllGather in a native
 * host, torch.distributed in the Python mirror).  Return 0 on success.  Two calls per BatchNorm layer and
 * step: 3*h floats forward (count, mean, M2 per column -> Chan combine in fixed rank order, so every rank
 * derives bit-identical statistics), 2*h floats backward (sum dy, sum dy*xhat).  dgamma/dbeta and all other
 * gradients stay LOCAL sums: the caller's gradient all-reduce completes them. */
typedef int (*glnn_exchange_fn)(void* ctx, const float* send, float* recv, int64_t floats, void* stream);

typedef struct glnn_mlp_step_desc {
  int32_t num_layers;
  int32_t batchnorm;      /* hidden-layer norm: 0 none, 1 nn.BatchNorm1d, 2 nn.LayerNorm (ABI 5: gamma/beta = its affine, bn_eps = its eps,
                           * mean[l]/rstd[l] = max_batch floats of per-ROW statistics, act[l] required, a_scale/a_shift/running_* unused) */
  int32_t dims[GLNN_MLP_MAX_LAYERS + 1];
  float dropout_p;
  float bn_eps;
  float bn_momentum;
  int64_t max_batch;
  float* w[GLNN_MLP_MAX_LAYERS];
  float* b[GLNN_MLP_MAX_LAYERS];
  float* gw[GLNN_MLP_MAX_LAYERS];
  float* gb[GLNN_MLP_MAX_LAYERS];
  float* gamma[GLNN_MLP_MAX_LAYERS];
  float* beta[GLNN_MLP_MAX_LAYERS];
  float* ggamma[GLNN_MLP_MAX_LAYERS];
  float* gbeta[GLNN_MLP_MAX_LAYERS];
  float* running_mean[GLNN_MLP_MAX_LAYERS];
  float* running_var[GLNN_MLP_MAX_LAYERS];
  int64_t* nbt[GLNN_MLP_MAX_LAYERS];
  float* mean[GLNN_MLP_MAX_LAYERS];
  float* rstd[GLNN_MLP_MAX_LAYERS];
  float* a_scale[GLNN_MLP_MAX_LAYERS];
  float* a_shift[GLNN_MLP_MAX_LAYERS];
  float* z[GLNN_MLP_MAX_LAYERS];
  int64_t ldz[GLNN_MLP_MAX_LAYERS];
  float* logits;
  int64_t ld_logits;
  float* dlogits;
  int64_t ld_dlogits;
  float* da;
  int64_t ld_da;
  float* dz;
  int64_t ld_dz;
  float* ws_bn;
  int64_t ws_bn_floats;
  float* ws_tn;
  int64_t ws_tn_floats;
  float* ws_gemm;
  int64_t ws_gemm_floats;
  float* ws_loss;
  int64_t ws_loss_floats;
  float* loss_out;
  float* loss_accum;
  /* batch split over ranks (all zero / NULL = single rank) */
  int32_t world;
  int32_t rank;
  glnn_exchange_fn exchange;
  void* exchange_ctx;
  float* sync_send;      /* >= 3 * max hidden floats */
  float* sync_recv;      /* >= world * 3 * max hidden floats */
  float* sync_rows;      /* 1 float: the global batch row count of the current step (written by the library) */
  /* optional: GLNN_MLP_COUNTERS device ints, ZERO when first handed over (the library leaves them zero).  With them the
   * BatchNorm statistics, the bias-gradient column sums and the loss finish inside their first launch (the last
   * workgroup folds the partials) instead of in a second one: 7 launches fewer per 3-layer step.  ws_loss must then hold
   * >= 256 * 65 floats for the last layer's bias gradient to ride along (label_dim <= 64).  NULL = two-launch forms. */
  int32_t* sync_counters;
  /* optional, per hidden layer l: act[l] [max_batch, ld_act[l]] receives dropout(relu(norm(z[l]))) once per step
   * (glnn_act_fwd_f32) and the next layer's forward and weight-gradient GEMMs read it as a plain operand.  NULL = that tail
   * is recomputed inside their operand loads (nothing stored): cheaper for narrow layers and small batches; with dropout
   * and a wide next layer the per-element mask hash, re-evaluated by every workgroup that stages the tile, costs more
   * than one 8-byte-per-element pass (MLP3w8: 316 vs 260+13 us forward, 309 vs 277 us weight gradient). */
  float* act[GLNN_MLP_MAX_LAYERS];
  int64_t ld_act[GLNN_MLP_MAX_LAYERS];
  /* optional: xb [max_batch, ld_xb] receives feats[idx] once per step (glnn_gather_rows_f32) and the first layer's forward and
   * weight-gradient GEMMs read it as a plain operand -- the weight gradient then qualifies for the pipelined kernel
   * (B=4096, 2048 x 100: 38 + 9 us gathered vs 29 us plain + a 4 us gather).  NULL (or idx == NULL) = rows gathered in the
   * operand loads. */
  float* xb;
  int64_t ld_xb;
  /* optional data-parallel hook: called on the host right after the weight-gradient GEMM of `layer` has been enqueued, i.e.
   * gw[layer] is complete in `stream` order while the rest of the backward (input gradient, BatchNorm backward, the layers
   * in front) is still to be issued -- the host can start that layer's gradient all-reduce on its own stream and overlap
   * it (glnn_amd.dist.OverlappedGradSync: MLP3w8's 16 MB hidden-layer gradient rides under ~0.35 ms of remaining backward).
   * Return 0 on success.  NULL = no call. */
  int (*grad_ready)(void* ctx, int layer, void* stream);
  void* grad_ready_ctx;
  /* optional second [max_batch, ld_dz2] gradient buffer (ABI 7: the aux_stream / ev_main / ev_aux fields of the two-stream backward
   * of ABI 5-6 are gone -- measured slower, removed): small steps collect their weight gradients and issue them as ONE launch at the
   * end of the backward; dz_l then alternates between dz and dz2 so that it outlives the loop. */
  float* dz2;
  int64_t ld_dz2;
} glnn_mlp_step_desc;

GLNN_API int glnn_mlp_fwd_bwd_f32(const glnn_mlp_step_desc* desc, const float* feats, int64_t ldx,
                                  const int64_t* idx, int64_t m, int kind, const int64_t* labels,
                                  const float* target_logp, int64_t ldt, const int64_t* target_rows,
                                  float lamb, const uint32_t* drop_seeds, void* stream);

/* ------------------------------------------------------------------------------------------
 * ABI 6: the WHOLE optimisation step of the student in one call -- glnn_mlp_fwd_bwd_f32 followed by glnn_adam_step_f32 on the
 * same stream (reference train_and_eval.py:74-85: forward, loss, backward, optimizer.step()), for hosts with nothing to put
 * between the two (no gradient exchange).  Because Adam is known to be the next launch and the only consumer of the gradients,
 * the backward leaves the final sums of its gradient partials to it (the split-K slabs of the weight gradients, the per-chunk
 * column sums behind the bias gradients, the per-workgroup loss partials): the fold launches and last-workgroup tails of the
 * two-call form disappear, the sums and their order do not (same bits), and the folded gradients are still stored to `grads`.
 * glnn_adam_desc: the arguments of glnn_adam_step_f32 plus `grads_host`, a HOST copy of the device pointer table `grads`
 * (num_tensors entries; how a pending fold finds its tensor).  num_tensors <= 32 for the folds; more tensors take the plain form.
 * ------------------------------------------------------------------------------------------ */
typedef struct glnn_adam_desc {
  float* const* params; const float* const* grads; float* const* exp_avg; float* const* exp_avg_sq; const int64_t* sizes;
  const float* const* grads_host;
  int32_t num_tensors; int32_t reserved;
  int64_t max_size;
  float lr, beta1, beta2, eps, weight_decay; int32_t reserved2;
  int64_t step;                       /* 1-based step count AFTER this update */
} glnn_adam_desc;

GLNN_API int glnn_mlp_train_step_f32(const glnn_mlp_step_desc* desc, const float* feats, int64_t ldx,
                                     const int64_t* idx, int64_t m, int kind, const int64_t* labels,
                                     const float* target_logp, int64_t ldt, const int64_t* target_rows,
                                     float lamb, const uint32_t* drop_seeds, const glnn_adam_desc* adam, void* stream);

/* ------------------------------------------------------------------------------------------
 * The whole forward + NLL + backward of ONE sampled-block GraphSAGE training step in one call (reference
 * train_and_eval.py:39-53 over SAGE.forward, models.py:101-119): per layer glnn_spmm_csr_f32 (SAGE_GCN) -> glnn_gemm_f32 ->
 * glnn_bn_stats_f32 -> glnn_act_fwd_f32; glnn_softmax_loss_f32 with labels[label_rows[i]]; backward per layer
 * glnn_gemm_tn_f32 -> glnn_gemm_f32 -> glnn_csr_transpose(add_self) + glnn_degrees_f32(1/(deg+1)) + glnn_spmm_csr_f32 (SUM
 * over the transposed block) -> glnn_bn_relu_bwd_f32.  Every buffer is caller-owned:
 *   layer[l].indptr/indices   block l (CSR over its destinations; layer[0] outermost), n_src of block l == n_dst of block l-1;
 *                             layer[0] may carry GLOBAL column ids with self_rows[v] = global id of destination v, x = feats
 *   w/b/gw/gb                 fc_neigh weight [dims[l+1], dims[l]] / bias and their gradients (contiguous)
 *   gamma..a_shift            BatchNorm1d after hidden layer l (batchnorm != 0)
 *   agg/z/h                   kept activations: aggregate [n_dst, dims[l]], pre-activation [n_dst, dims[l+1]] (z of the last
 *                             layer = logits), hidden output [n_dst, dims[l+1]] (float4 rows: leading dims % 4 == 0)
 *   t_indptr/t_indices/inv_deg/tr_ws  (l >= 1) outputs / workspace of the block's transpose (glnn_csr_transpose sizes)
 *   dagg [max n_dst_l, max dims[l]], dh [max n_src_l, max dims[l]] (l >= 1)   backward scratch
 * Round 5 (same ABI, same results to rounding): with batchnorm != 0 and float4-addressable buffers big enough (ws_bn >= 2 slots dims[1] +
 * 5 dims[1] + 8 floats, slots = n_dst_0 / 128 + <= 512; ws_tn >= 8 dims[0] dims[1] floats) the BatchNorm backward of the OUTERMOST layer has
 * no pass of its own -- the transposed aggregation stores dy = da behind the tail's masks and the column sums, dz = alpha dy + beta z +
 * gamma is evaluated in the operand loads of dW_0; gb of layer 0 (true gradient 0 in front of a BatchNorm) is then exactly 0.  Otherwise,
 * or with GLNN_SAGE_FUSE_BN_APPLY=0 / GLNN_SAGE_FUSE_BN_DY=0, the launch sequence above.  Blocks of <= 6 in-edges per row on average
 * (layer[l].nnz) take the four-rows-per-wave aggregation kernel (GLNN_SPMM_SHORT=0: never; same bits).
 * The optimiser step is NOT included: call glnn_adam_step_f32 next.
 * ------------------------------------------------------------------------------------------ */
#define GLNN_SAGE_MAX_LAYERS 8
typedef struct glnn_sage_layer {
  const int64_t* indptr; const int32_t* indices; int64_t n_dst, n_src, nnz;
  const int64_t* self_rows;
  float* w; float* b; float* gw; float* gb;
  float* gamma; float* beta; float* ggamma; float* gbeta; float* running_mean; float* running_var; int64_t* nbt;
  float* mean; float* rstd; float* a_scale; float* a_shift;
  float* agg; int64_t ld_agg; float* z; int64_t ldz; float* h; int64_t ldh;      /* h may be NULL (hidden layers): the tail of z is then
                                                                                   * applied inside the next layer's aggregation, never written */
  int64_t* t_indptr; int32_t* t_indices; float* inv_deg; void* tr_ws; int64_t tr_ws_bytes;
  /* (layers >= 1) scratch the step fills: the block transposed with one self entry per destination (glnn_csr_transpose(..., add_self = 1)) and
   * 1 / (in-degree + 1) (glnn_degrees_f32, GLNN_DEG_INV_PLUS1).  tr_ws == NULL: t_indptr / t_indices / inv_deg already HOLD them -- built by
   * the caller, e.g. by its batch loader beside the previous step (round 5: the ten launches leave the step's stream). */
  uint32_t drop_seed;
} glnn_sage_layer;

typedef struct glnn_sage_step_desc {
  int32_t num_layers;
  int32_t batchnorm;
  int32_t dims[GLNN_SAGE_MAX_LAYERS + 1];
  float dropout_p, bn_eps, bn_momentum, lamb;
  glnn_sage_layer layer[GLNN_SAGE_MAX_LAYERS];
  const float* x; int64_t ldx; int64_t x_rows;        /* source matrix of layer 0: feats (global ids) or the gathered batch features */
  const int64_t* labels; const int64_t* label_rows;   /* labels[label_rows[i]] for destination i of the last block (label_rows NULL = i) */
  float* dlogits; int64_t ld_dlogits;
  float* dagg; int64_t ld_dagg; float* dh; int64_t ld_dh;
  float* ws_bn; int64_t ws_bn_floats; float* ws_tn; int64_t ws_tn_floats; float* ws_gemm; int64_t ws_gemm_floats;
  float* ws_loss; int64_t ws_loss_floats;
  float* loss_out; float* loss_accum;
} glnn_sage_step_desc;

GLNN_API int glnn_sage_fwd_bwd_f32(const glnn_sage_step_desc* desc, void* stream);
/* ... and the same followed by the fused Adam launch in ONE call (ABI 11; reference train_and_eval.py:39-53 incl. optimizer.step()):
 * the backward leaves its last partial sums (split slabs of the weight gradients, the last layer's bias-gradient column partials, the
 * per-workgroup losses) to Adam -- the same parameters, moments and loss as glnn_sage_fwd_bwd_f32 + glnn_adam_step_f32, bit for bit.
 * `adam` as in glnn_mlp_train_step_f32. */
GLNN_API int glnn_sage_train_step_f32(const glnn_sage_step_desc* desc, const glnn_adam_desc* adam, void* stream);
/* ABI 10: ws_bn floats with which the outermost layer's BatchNorm backward takes the form without passes of its own (see above) for an
 * outermost block of n_dst_0 destinations and dims[1] = hidden; the step's other uses of ws_bn need (3 chunks + 2 + 3 ceil(chunks / 64)) *
 * max hidden + 1024 floats, chunks = ceil(max n_dst / 128): allocate the larger of the two. */
GLNN_API int64_t glnn_sage_step_ws_bn_floats(int64_t n_dst_0, int hidden);

/* y = dropout(relu(z * a_scale + a_shift)) materialised (a_scale/a_shift NULL: plain ReLU): the `norms[l](h)` ->
 * `activation` -> `dropout` tail of a TRAINING-mode SAGE layer (reference models.py:113-117), whose output the next
 * layer's aggregation gathers.  a_scale/a_shift come from glnn_bn_stats_f32; the backward is glnn_bn_relu_bwd_f32 with
 * the same (drop_p, drop_seed).  z, y: float4 rows (leading dimensions % 4 == 0, >= round4(h)); padding columns of y = 0. */
GLNN_API int glnn_act_fwd_f32(const float* z, int64_t ldz, int64_t rows, int h, const float* a_scale,
                              const float* a_shift, float drop_p, uint32_t drop_seed, float* y, int64_t ldy,
                              void* stream);

/* glnn_act_fwd_f32 / glnn_bn_relu_bwd_f32 with the tail's ReLU optional (ABI 5).  relu = 0 is the hidden-layer tail of the
 * reference's GCN (models.py:189-199: GraphConv(activation=relu) -> norms[l] -> dropout -- the ReLU sits inside the conv, IN
 * FRONT of the norm; train.conf.yaml's pokec / penn94 GCN sections use norm_type batch): y = dropout(z * a_scale + a_shift),
 * and the backward gates nothing by the sign of the normalised value. */
GLNN_API int glnn_norm_drop_fwd_f32(const float* z, int64_t ldz, int64_t rows, int h, const float* a_scale,
                                    const float* a_shift, int relu, float drop_p, uint32_t drop_seed, float* y,
                                    int64_t ldy, void* stream);
GLNN_API int glnn_bn_bwd_f32(const float* da, int64_t ldda, const float* z, int64_t ldz, int64_t rows, int h,
                             const float* gamma, const float* mean, const float* rstd, const float* a_scale,
                             const float* a_shift, int relu, float drop_p, uint32_t drop_seed, float* dz,
                             int64_t lddz, float* dgamma, float* dbeta, float* dz_col_sum, float* workspace,
                             int64_t workspace_floats, void* stream);

/* nn.LayerNorm(hidden) as a hidden-layer tail (reference models.py:28-31, 87-90, 174-186; train.conf.yaml uses norm_type
 * "layer" for the house_class MLP):
 *   forward   y = dropout(relu?(xhat * gamma + beta)),  xhat = (z - mean_row) * rstd_row,  rstd = 1/sqrt(biased var + eps);
 *             mean_out / rstd_out [rows] (optional) are what the backward needs; gamma/beta NULL = no affine.
 *   backward  dy = da * keep/(1-p) * [relu ? xhat*gamma+beta > 0 : 1];  dgamma = sum_rows dy*xhat, dbeta = sum_rows dy;
 *             dz = rstd * (dy*gamma - mean_cols(dy*gamma) - xhat * mean_cols(dy*gamma*xhat));  dz_col_sum (optional) = sum_rows dz
 *             = the bias gradient of the Linear in front.  workspace >= glnn_layernorm_bwd_workspace_floats(rows, h) floats when
 *             any column sum is requested; h <= 4096.  In place on da allowed (dz == da).
 * Rows are float4 rows (leading dimensions % 4 == 0, >= round4(h), 16-byte aligned); padding columns of y / dz = 0. */
GLNN_API int glnn_layernorm_fwd_f32(const float* z, int64_t ldz, int64_t rows, int h, const float* gamma,
                                    const float* beta, float eps, int relu, float drop_p, uint32_t drop_seed,
                                    float* y, int64_t ldy, float* mean_out, float* rstd_out, void* stream);
GLNN_API int64_t glnn_layernorm_bwd_workspace_floats(int64_t rows, int h);
GLNN_API int glnn_layernorm_bwd_f32(const float* da, int64_t ldda, const float* z, int64_t ldz, int64_t rows, int h,
                                    const float* gamma, const float* beta, const float* mean, const float* rstd,
                                    int relu, float drop_p, uint32_t drop_seed, float* dz, int64_t lddz,
                                    float* dgamma, float* dbeta, float* dz_col_sum, float* workspace,
                                    int64_t workspace_floats, void* stream);

/* The dropout keep-mask the kernels above evaluate on the fly (nn.Dropout, reference models.py:52):
 * mask[r*h + c] = 1 if element (r,c) is kept under (drop_p, drop_seed).  torch's Philox stream cannot
 * be reproduced by a custom kernel, so parity tests run the oracle with THIS mask as an input. */
GLNN_API int glnn_dropout_mask_u8(int64_t rows, int h, float drop_p, uint32_t drop_seed,
                                  uint8_t* mask, void* stream);

/* Uniform in-neighbour sampling WITHOUT replacement for a list of seed (destination) rows: at most `fanout`
 * sources per seed, all of them when in_deg <= fanout (dgl semantics).  Replaces the CPU-side
 * dgl.dataloading.MultiLayerNeighborSampler of reference train_and_eval.py:179-190 (one call per layer).
 * out_src [n_seeds, fanout] int32 global source ids (first out_cnt[i] entries of row i valid). */
GLNN_API int glnn_sample_neighbors(const int64_t* indptr, const int32_t* indices, const int64_t* seeds,
                                   int64_t n_seeds, int fanout, uint32_t rng_seed, int32_t* out_src,
                                   int32_t* out_cnt, void* stream);

/* ------------------------------------------------------------------------------------------
 * Block construction for mini-batch teacher training / chunked inference: what dgl.dataloading.NodeDataLoader does on
 * the CPU per batch in the reference (train_and_eval.py:176-205; blocks consumed at :41 and models.py:109,134-137),
 * here on the device and sized by the batch's FRONTIER (hash table + two single-pass scans), never by N.
 *   seeds [ns]        the block's destination nodes (global ids)
 *   sampled mode      smp_src [ns, fanout] / smp_cnt [ns] from glnn_sample_neighbors, nnz_cap >= ns * fanout
 *   full mode         smp_src = smp_cnt = NULL: every in-edge of every seed from (g_indptr, g_indices);
 *                     nnz_cap >= the block's edge count (the caller's bound; counts[0] reports the true number)
 * Outputs (caller-owned): indptr [ns+1]; indices [nnz_cap] LOCAL source ids (block CSR over its destinations, edges in
 * the order of smp_src / of the graph's rows); gindices [nnz_cap] the same edges' GLOBAL source ids (optional);
 * input_nodes [ns + nnz_cap]: the block's source nodes = its destinations first (models.py:109), then every other
 * source in order of first appearance; counts (device int64[2]) = {nnz, number of source nodes}.
 * workspace: glnn_block_workspace_bytes(ns, nnz_cap) bytes, 8-byte aligned.  Node ids must be < 0x7F7F7F7F.
 * GLOBAL-ID BLOCK (round 5): indices == NULL and input_nodes == NULL with gindices != NULL -- only indptr and the edges' global source ids
 * are produced (row-count scan + one copy pass; no table, no relabelling), counts = {nnz, -1}: the outermost block of a training batch,
 * whose consumer (glnn_sage_fwd_bwd_f32 with self_rows) gathers from the global feature matrix and never reads local ids. */
GLNN_API int64_t glnn_block_workspace_bytes(int64_t ns, int64_t nnz_cap);
GLNN_API int glnn_block_build(const int64_t* g_indptr, const int32_t* g_indices, const int64_t* seeds,
                              int64_t ns, const int32_t* smp_src, const int32_t* smp_cnt, int fanout,
                              int64_t nnz_cap, int64_t* indptr, int32_t* indices, int32_t* gindices,
                              int64_t* input_nodes, int64_t* counts, void* workspace,
                              int64_t workspace_bytes, void* stream);
/* ABI 8: the same, told the size of the id universe (every seed and neighbour id is < n_nodes; 0 = unknown -> glnn_block_build).
 * When n_nodes is no larger than the two arrays of the frontier hash table (the wider blocks of a large batch), the positions are indexed by
 * the node id itself: one atomicMin per edge, no probing, a smaller fill.  Same workspace size, identical results.  Precondition: every id is in
 * [0, n_nodes); an id outside is never used as an index -- counts[0] comes back as -1 (the block's outputs are then undefined). */
GLNN_API int glnn_block_build_ids(const int64_t* g_indptr, const int32_t* g_indices, const int64_t* seeds,
                                  int64_t ns, const int32_t* smp_src, const int32_t* smp_cnt, int fanout,
                                  int64_t nnz_cap, int64_t* indptr, int32_t* indices, int32_t* gindices,
                                  int64_t* input_nodes, int64_t* counts, int64_t n_nodes, void* workspace,
                                  int64_t workspace_bytes, void* stream);

/* Transposed CSR (rows = the original SOURCE nodes, entries = destinations, sorted within a row): the graph
 * loss.backward() walks through dgl's SpMM in the reference (train_and_eval.py:27,54) -- A^T dY is then the same
 * glnn_spmm_csr_f32 gather on the result, deterministic, no float atomics:
 *     d/dx of SUM(row_scale, col_scale)  =  SUM over the transposed CSR with row_scale <-> col_scale swapped;
 *     d/dx of SAGE_GCN                   =  SUM over the transposed CSR built with add_self != 0 (one extra entry u <- u
 *                                           for every destination u < n_dst: the h_dst term) and col_scale = 1/(deg+1).
 * t_indptr [n_src+1], t_indices [nnz + (add_self ? n_dst : 0)]; nnz = indptr[n_dst] (host value);
 * workspace: glnn_csr_transpose_workspace_bytes(n_src, nnz + (add_self ? n_dst : 0)) bytes, 8-byte aligned. */
GLNN_API int64_t glnn_csr_transpose_workspace_bytes(int64_t n_src, int64_t nnz_out);
GLNN_API int glnn_csr_transpose(const int64_t* indptr, const int32_t* indices, int64_t n_dst, int64_t n_src,
                                int64_t nnz, int add_self, int64_t* t_indptr, int32_t* t_indices,
                                void* workspace, int64_t workspace_bytes, void* stream);

/* K7  row gather: out[i,:] = x[rows[i],:]  (feats[idx], reference train_and_eval.py:42,76,
 *     models.py:136) and scatter y[rows[i],:] = x[i,:] (models.py:145). */
GLNN_API int glnn_gather_rows_f32(const float* x, int64_t ldx, const int64_t* rows, int64_t n_rows,
                                  int d, float* out, int64_t ldo, void* stream);
GLNN_API int glnn_scatter_rows_f32(const float* x, int64_t ldx, const int64_t* rows, int64_t n_rows,
                                   int d, float* out, int64_t ldo, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GLNN_HIP_H */
