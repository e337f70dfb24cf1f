// Shared host-side helpers for libglnn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "glnn_hip.h"

namespace glnn {

struct GradFold;
struct PendingFolds;
struct BnApplyA;

void set_error(const char* fmt, ...);

// Process-wide switches between SUPPORTED forms of the same arithmetic (every form is held to the same results by the tests: the
// switches exist for those tests and for A/B timing).  Read from the environment ONCE, at the first call into the library
// (capi.hip: the library's only getenv); glnn_reload_options() re-reads them (tests flip a switch between two runs in one process).
// All default to the shipping form.
struct Options {
  int gemm_pipe;            // GLNN_GEMM_PIPE=0: every GEMM on the compiler-scheduled kernels (no hand-scheduled main loop)
  int gemm_rowpanel;        // GLNN_GEMM_ROWPANEL=0: short reductions stay on the tiled kernels
  int gemm_lat;             // GLNN_GEMM_LAT=0: small batches keep the tiled GEMM + separate reduction launches (mlp_lat.hip off)
  int gemm_tn_lat;          // GLNN_GEMM_TN_LAT=0: the small step's weight gradients as the batched tiled launch
  int gemm_tn_lat_splits;   // GLNN_GEMM_TN_LAT_SPLITS (8): 1 = unsplit reduction (bit-identical to the two-call step)
  int lat_bn_bwd;           // GLNN_STUDENT_LAT_BN_BWD=0: input gradient and BatchNorm backward as separate tiled launches
  int bn_bwd_one_launch;    // GLNN_BN_BWD_ONE_LAUNCH=0: BatchNorm backward as partial + apply launches
  int narrow_bwd;           // GLNN_STUDENT_NARROW_BWD=0: the classifier's input gradient is always written
  int64_t narrow_bwd_min;   // GLNN_STUDENT_NARROW_BWD_MIN (1 << 20): rows x hidden width from which it is recomputed instead
  int narrow_wgrad;         // GLNN_STUDENT_NARROW_WGRAD=0: the classifier's weight gradient stays a gemm_tn launch
  int pad_w0;               // GLNN_STUDENT_PAD_W0=0: wide unaligned first layers stay on the unaligned-W latency kernel
  int slab_consumers;       // GLNN_STUDENT_SLAB_CONSUMERS=0: split-K partials are folded by a launch, not by their consumer
  int defer_stats;          // GLNN_STUDENT_DEFER_STATS=0: small-step BatchNorm statistics finalised by the producing launch
  int batched_wgrad;        // GLNN_STUDENT_BATCHED_WGRAD=0: small-step weight gradients one launch per layer
  int fuse_apply;           // GLNN_STUDENT_FUSE_APPLY=0: the first hidden layer's BatchNorm-backward apply stays its own launch
  int adam_folds;           // GLNN_STUDENT_ADAM_FOLDS=0: gradient partials are folded before Adam, not by it
  int gemm_stats;           // GLNN_GEMM_STATS=0: BatchNorm statistics always take their own first pass over the GEMM's output
  int spmm_short;           // GLNN_SPMM_SHORT=0: sparse training blocks stay on the one-row-per-wave aggregation kernel
  int sage_fuse_bn_dy;      // GLNN_SAGE_FUSE_BN_DY=0: the deferred BatchNorm backward of layer 0 keeps its dy pass (the transposed aggregation writes da)
  int sage_fuse_bn_apply;   // GLNN_SAGE_FUSE_BN_APPLY=0: teacher training writes layer 0's dz (BatchNorm-backward apply as its own launch)
  int bn0_consts_in_gemm;   // GLNN_STUDENT_BN0_CONSTS_IN_GEMM=0: the constants of the deferred apply come from a launch of their own (bn_bwd_parts_finish)
  int bn0_in_gemm;          // GLNN_STUDENT_BN0_IN_GEMM=0: the first hidden layer's BatchNorm backward stays partial + apply launches behind a plain input-gradient GEMM
  int signal_fence;         // GLNN_SIGNAL_NO_FENCE=1: glnn_stream_wait_value32 without the empty kernel behind the wait (tests: the negative control)
  int cls_fused;            // GLNN_STUDENT_CLS_FUSED=0: the large-batch classifier stays a split-K GEMM launch + a loss launch (cls_block.hip off)
};
const Options& opts();

inline int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
inline int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  set_error("%s", buf);
  return code;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(GLNN_ERR_HIP, "%s: launch failed: %s", what, hipGetErrorString(e));
  return GLNN_OK;
}

__host__ __device__ inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#define GLNN_REQUIRE(cond, ...)                                     \
  do {                                                              \
    if (!(cond)) return ::glnn::fail(GLNN_ERR_INVALID_ARG, __VA_ARGS__); \
  } while (0)

constexpr int kWave = 64;  // CDNA wavefront

// Counter-based dropout: keep(row, col) under (seed, p).  One lowbias32 integer hash per PAIR of adjacent columns,
// 16 bits of it per element (p is quantised to 1/65536); the same function is evaluated by the forward operand
// transform, the weight-gradient operand transform and the activation backward, so no mask is ever stored.
__host__ __device__ inline uint32_t drop_hash(uint32_t seed, uint32_t row, uint32_t col) {
  uint32_t h = seed ^ (row * 0x9E3779B1u) ^ (col * 0x85EBCA77u + 0x632BE5ABu);
  h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
  return h;
}
// dz of one element of a BatchNorm backward:  gamma*rstd * (dy - S1/B - xhat * S2/B),  xhat = (z - mean) * rstd.  ONE definition with the
// contraction pinned (one fma, chosen here), because five kernels evaluate it and several pairs of them are held to identical bits.
__device__ __forceinline__ float bn_dz(float grs, float dy, float c1, float z, float mu, float rs, float c2) {
#pragma clang fp contract(off)
  const float xh = (z - mu) * rs;
  const float t = dy - c1;
  return grs * fmaf(-xh, c2, t);
}
__host__ __device__ inline uint32_t drop_threshold(float p) {   // keep iff the element's 16 bits >= threshold
  return (uint32_t)(p * 65536.0f);
}
__host__ __device__ inline bool drop_keep(uint32_t seed, uint32_t thr, uint32_t row, uint32_t col) {
  const uint32_t h = drop_hash(seed, row, col >> 1);
  return ((col & 1u) ? (h >> 16) : (h & 0xFFFFu)) >= thr;
}

// The batch-split group of glnn_mlp_step_desc as the BatchNorm kernels see it (student.hip); NULL = single rank.
struct BnGroup {
  int world, rank;
  glnn_exchange_fn exchange;
  void* ctx;
  float* send;
  float* recv;
  float* rows_out;   // device float: global row count of the step (written in the forward, read in the backward)
};
// `counters` (optional, device ints, all zero on entry, left zero on exit): with them the final fold of a two-stage reduction
// is done by the last workgroup of the first stage instead of by a second launch (student.hip: last_workgroup()).
struct ColStats;
int bn_stats(const float* z, int64_t ldz, int64_t rows, int h, const float* gamma, const float* beta, float eps, float momentum,
             float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean_out, float* rstd_out,
             float* a_scale_out, float* a_shift_out, float* workspace, int64_t workspace_floats, void* stream, const BnGroup* g,
             int* counters = nullptr, const float* slabs = nullptr, int nslab = 0, const float* bias = nullptr, const ColStats* cs = nullptr);
// cs (cs->done): the producing GEMM left the first-pass partials in `workspace` (glnn::gemm_stats): only the combine is launched
// slabs / nslab (<= 8) / bias: z = sum of the split-K partial slabs of gemm_split_partials + bias, folded (and stored to z) by the
// statistics kernel itself; one-launch form only (counters, single rank) -- otherwise GLNN_ERR_UNSUPPORTED with nothing launched
int softmax_loss(const float* logits, int64_t ldz, int64_t rows, int c, int kind, const int64_t* labels, const int64_t* label_rows,
                 const float* target_logp, int64_t ldt, const int64_t* target_rows, float lamb, float* dlogits, int64_t ldg,
                 float* logprob_out, int64_t ldl, float* loss_out, float* loss_accum, float* workspace, int64_t workspace_floats,
                 void* stream, int* counter, float* col_sum, const float* slabs = nullptr, int nslab = 0, const float* bias = nullptr,
                 struct PendingFolds* pf = nullptr);
// pf: the loss / bias-gradient partials are registered there for the fused Adam launch instead of being folded by the last workgroup
// student.hip: loss = sum of n per-workgroup partials * inv_rows (the second launch of the counter-less loss form)
int loss_finalize(const float* partial, int n, float inv_rows, float* loss_out, float* loss_accum, void* stream);
// cls_block.hip (round 6): logits = tail(a) . w^T + bias of a LARGE batch in front of a narrow classifier (n <= 48, k % 256 == 0, k <= 4096) by
// row-local workgroups, with log_softmax + loss + dlogits behind it in the same launch (ls != NULL; softmax_loss's counter-less form:
// per-four-rows loss partials in ls->ws, folded by the fused Adam launch (ls->pf) or by loss_finalize).  a_scale / a_shift NULL: a is the
// stored tail.  GLNN_ERR_UNSUPPORTED = nothing launched.
struct ClsLoss {
  int kind; const int64_t* labels; const int64_t* label_rows; const float* target_logp; int64_t ldt; const int64_t* target_rows; float lamb;
  float* dlogits; int64_t ldg; float* loss_out; float* loss_accum; float* ws; int64_t ws_floats; struct PendingFolds* pf;
};
int cls_fwd(const float* a, int64_t lda, const float* a_scale, const float* a_shift, float drop_p, uint32_t drop_seed, int64_t m, int k,
            const float* w, int64_t ldw, int n, const float* bias, float* logits, int64_t ldz, const ClsLoss* ls, void* stream);
// slabs / nslab / bias: the logits are still the split-K partials of gemm_split_partials (slabs[s][rows][c]); they are summed, the
// bias added and the result stored to `logits` by the loss kernel itself
// da = dl[rows, k] . w[k, h] (w rows ldw apart: a Linear's [out = k, in = h] weight), k <= 64: see bn_bwd_*_sk in student.hip
struct NarrowProduct {
  const float* dl; int64_t lddl; int k; const float* w; int64_t ldw;
  // optional (both or dw_ws alone): the first pass also leaves the narrow layer's OWN gradients as row-chunk partials -- dw_ws[chunk][k][h]
  // (dW = dl^T . act(z), 16-byte aligned) and db_ws[chunk][k] (column sums of dl), chunk < ceil(rows / 128): fold k ascending
  float* dw_ws; float* db_ws;
};
int bn_relu_bwd(const float* da, int64_t ldda, const float* z, int64_t ldz, int64_t rows, int h, const float* gamma,
                const float* mean, const float* rstd, const float* a_scale, const float* a_shift, float drop_p, uint32_t drop_seed,
                float* dz, int64_t lddz, float* dgamma, float* dbeta, float* dz_col_sum, float* workspace, int64_t workspace_floats,
                void* stream, const BnGroup* g, int* counters = nullptr, int relu = 1, int da_slabs = 0,
                struct GradFold* defer_colsum = nullptr, const NarrowProduct* prod = nullptr, struct BnApplyA* defer_apply = nullptr);
// defer_apply (round 5): dz is NOT written.  One pass leaves dy (da behind the dropout and ReLU masks) IN PLACE of da and the per-chunk sums
// (the partial pass's own numbers: dgamma / dbeta are the plain form's bits), a constants launch turns the totals into alpha / beta / gamma
// with dz = alpha dy + beta z + gamma, and *defer_apply describes them for the ONE consumer of dz (gemm_tn(..., bn)).  dz_col_sum (the bias
// gradient in front of the BatchNorm, mathematically 0) is written as 0.  GLNN_ERR_UNSUPPORTED with nothing launched unless: BatchNorm,
// single rank, da in memory and writable (da == dz: in place), more than 64 row chunks, h % 4 == 0, float4-addressable rows.
// prod: the input gradient da is NOT in memory (da / ldda ignored): both passes recompute da = dl . w on the matrix cores; BatchNorm
// two-launch form only (GLNN_ERR_UNSUPPORTED with nothing launched otherwise)
// defer_colsum: (one-launch form only) dz_col_sum is NOT written; *defer_colsum describes the per-chunk partials left in `workspace`
// da_slabs > 1: da points at that many split-K partial slabs (da[s][rows][ldda]) which the one-launch form sums itself; any other
// form returns GLNN_ERR_UNSUPPORTED with nothing launched (fold with gemm_fold_partials, call again)
int gemm_fold_partials(const float* workspace, int splits, int64_t m, int n, const float* bias, float* c, int64_t ldc, void* stream);
// student.hip: dst[r][0:ldd] = src[r][0:cols] followed by zeros (ldd % 4 == 0, dst 16-byte aligned): a float4-addressable copy of a matrix
int pad_rows(const float* src, int64_t lds, int64_t rows, int cols, float* dst, int64_t ldd, void* stream);
// student.hip: out[0:h] = sum over k < nchunks of ws[k][0:h], k ascending (the order in which the fused Adam launch folds a GradFold with lanes4 = 0)
int chunk_sum(const float* ws, int nchunks, int h, float* out, void* stream);

// gemm.hip: several independent weight gradients (arguments as glnn_gemm_tn_f32, no column sums) in one gemm + one fold launch;
// GLNN_ERR_UNSUPPORTED (nothing launched) when a problem does not qualify -- see gemm_tn_batch
struct TnProblem {
  const float* a; int64_t lda; int64_t m; int ka; const float* b; int64_t ldb; const int64_t* b_rows; const float* b_scale;
  const float* b_shift; float drop_p; uint32_t drop_seed; int nb; float* c; int64_t ldc;
  // optional, gemm_tn_lat only (zero otherwise): `a` is dy and A = the BatchNorm backward's apply of (dy, bn_z) with the S1 / S2 tile partials
  // bn_p1 / bn_p2 [bn_nparts][ka] -- see TnLatProblem; bn_colsum = the bias-gradient tensor whose column sums are left to Adam
  const float* bn_z; int64_t bn_ldz; const float* bn_gamma; const float* bn_mean; const float* bn_rstd; const float* bn_p1; const float* bn_p2;
  int bn_nparts; float* bn_dgamma; float* bn_dbeta; float* bn_colsum;
};
// A gradient tensor whose final sum has been left to its consumer, the fused Adam launch (the fold launches / last-workgroup tails
// that used to produce it are gone): grad[i] = sum over k < nslab of src[k * stride + i], k ascending (lanes4 = 0) or in four
// interleaved lanes k = r, r + 4, ... combined as (l0 + l1) + (l2 + l3) (lanes4 = 1: the order of the column-sum folds).
struct GradFold { const float* grad; const float* src; int nslab; int lanes4; int64_t stride; };
struct LossFoldJob { const float* partial; int nblocks; float inv_rows; float* loss_out; float* loss_accum; };
constexpr int kMaxGradFolds = 40;
struct PendingFolds { int n; GradFold e[kMaxGradFolds]; int has_loss; LossFoldJob loss; };
// defer (n entries, optional): the fold launch is skipped; defer[p] describes problem p's slabs (nslab = 0: c was written directly)
int gemm_tn_batch(const TnProblem* problems, int n, float* workspace, int64_t workspace_floats, void* stream, GradFold* defer = nullptr);
// gemm.hip: glnn_gemm_tn_f32 with its folds optionally left to the fused Adam launch -- see the definition
// bn (optional, round 5): the operand is NOT a but dz = alpha[col] * a + beta[col] * z + gamma[col] -- the BatchNorm backward's apply written
// as an affine map of (dy, z): `a` holds dy (the upstream gradient behind the tail's dropout and ReLU masks, left in place by
// bn_relu_bwd(..., defer_apply)) and dz is never written.  Evaluated on the staged operand pieces of the pipelined kernel;
// GLNN_ERR_UNSUPPORTED (nothing launched) for any other shape -- ask gemm_tn_takes_bn first.
struct BnApplyA { const float* z; int64_t ldz; const float* alpha; const float* beta; const float* gamma;
                  // (round 6) p1 != NULL instead of alpha / beta / gamma: the per-tile column sums S1 / S2 [nparts][ka] are still unfolded; the product
                  // folds them (ascending) and makes the constants in its prologue, and the workgroups of the first column stripe store
                  // dgamma = S2, dbeta = S1 (and colsum = 0) -- bn_bwd_parts_finish's results without its launch
                  const float* p1 = nullptr; const float* p2 = nullptr; int nparts = 0; int64_t rows = 0; const float* bn_gamma = nullptr;
                  const float* bn_mean = nullptr; const float* bn_rstd = nullptr; float* dgamma = nullptr; float* dbeta = nullptr; float* colsum = nullptr; };
bool gemm_tn_takes_bn(const float* a, int64_t lda, int64_t m, int ka, const float* b, int64_t ldb, int nb, const float* z, int64_t ldz,
                      int64_t workspace_floats = -1);      // workspace_floats >= 0: also the split-dependent window condition, planned against that workspace
int gemm_tn(const float* a, int64_t lda, int64_t m, int ka, const float* b, int64_t ldb, const int64_t* b_rows, const float* b_scale,
            const float* b_shift, float drop_p, uint32_t drop_seed, int nb, float* c, int64_t ldc, float* col_sum_a, float* workspace,
            int64_t workspace_floats, void* stream, GradFold* defer, GradFold* defer_colsum, int64_t* used_floats, int64_t plan_floats = 0,
            const BnApplyA* bn = nullptr);
// student.hip: glnn_adam_step_f32 whose gradient reads fold the pending partial sums (and store the folded gradient); grads_host =
// host copy of the `grads` pointer table (how a pending fold finds its tensor); pending may be NULL
int adam_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq, const int64_t* sizes,
              int num_tensors, int64_t max_size, float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
              const float* const* grads_host, const PendingFolds* pending, void* stream);
// gemm.hip: split-K partials only (the consumer folds) -- see the definition
int gemm_split_partials(const float* a, int64_t lda, const int64_t* a_rows, const float* a_scale, const float* a_shift, float drop_p,
                        uint32_t drop_seed, int64_t m, int k, const float* b, int64_t ldb, int b_layout, int n, float* workspace,
                        int64_t workspace_floats, int* splits, void* stream);

// spmm.hip: the SAGE-"gcn" aggregation over rows stored as pre-activations z, the hidden layer's tail applied in the gather
struct SourceTail { const float* scale; const float* shift; float drop_p; uint32_t drop_seed; };      // scale / shift NULL: no affine (norm "none")
// nnz (optional): the block's edge count; a sparse block (<= 6 in-edges per row on average; plain SAGE aggregation) takes the short-row kernel (same bits)
int spmm_csr_tail(const int64_t* indptr, const int32_t* indices, int64_t n_dst, int64_t n_src, const float* z, int64_t ldz, int d,
                  const SourceTail& tail, float* out, int64_t ldo, void* stream, int64_t nnz = -1);
// spmm.hip: the transposed aggregation A^T dY (col-scaled SUM) whose epilogue is the first pass of the BatchNorm backward of the layer in
// front: out = dy (da behind the tail's dropout / ReLU masks -- da itself is never written), per-workgroup column sums S1 / S2 in `ws`
// (*nslots slots; student.hip's bn_bwd_deferred_finish folds them).  GLNN_ERR_UNSUPPORTED = nothing launched.
struct BnTail { const float* z; int64_t ldz; const float* mean; const float* rstd; const float* a_scale; const float* a_shift; float drop_p;
                uint32_t drop_seed; int relu; };
int spmm_csr_bn_dy(const int64_t* indptr, const int32_t* indices, int64_t n_dst, int64_t n_src, const float* x, int64_t ldx, int d,
                   const float* col_scale, const BnTail& tail, float* out, int64_t ldo, float* ws, int64_t ws_floats, int* nslots, void* stream);
// gemm.hip (round 6): the input-gradient product of a hidden layer whose epilogue is the first pass of that layer's BatchNorm backward --
// see the definition; student.hip's bn_bwd_parts_finish turns the tile partials into the constants of the deferred apply
int gemm_bn_dy(const float* a, int64_t lda, int64_t m, int k, const float* w, int64_t ldw, int n, const BnTail& t, float* c, int64_t ldc, float* s1,
               float* s2, void* stream);
// student.hip: S1 / S2 = sums over nparts partials (ascending), then dz = alpha dy + beta z + gamma as *defer_apply (cst: 3 h floats,
// 16-byte aligned), dgamma = S2, dbeta = S1, dz_col_sum = 0 -- ONE launch
int bn_bwd_parts_finish(const float* s1, const float* s2, int nparts, int h, int64_t rows, const float* z, int64_t ldz, const float* gamma,
                        const float* mean, const float* rstd, float* cst, float* dgamma, float* dbeta, float* dz_col_sum,
                        struct BnApplyA* defer_apply, void* stream);
// student.hip: the rest of bn_relu_bwd(..., defer_apply) behind partial sums that already exist (ws1 = ws, ws2 = ws + nslots h): fold,
// constants, *defer_apply.  ws must hold 2 nslots h + 5 h + 8 floats.
int bn_bwd_deferred_finish(float* ws, int64_t ws_floats, int nslots, int h, int64_t rows, const float* z, int64_t ldz, const float* gamma,
                           const float* mean, const float* rstd, float* dgamma, float* dbeta, float* dz_col_sum, struct BnApplyA* defer_apply,
                           void* stream);
int spmm_csr_nnz(const int64_t* indptr, const int32_t* indices, int64_t n_dst, int64_t n_src, int64_t nnz, const float* x, int64_t ldx, int d,
                 int mode, const float* col_scale, const float* x_self, int64_t ld_self, const int64_t* self_rows, float* out, int64_t ldo,
                 void* stream);

// gemm_rowpanel.hip: C = epi(A . W^T) for short reductions (k <= 128) over many rows -- persistent workgroups that keep a 128-column
// panel of W in LDS and walk row tiles of A; bit-identical to the tiled kernels.  GLNN_ERR_UNSUPPORTED = nothing launched.
// BatchNorm statistics partials out of a GEMM's epilogue (round 4): the product kernel leaves, per column of C, partial (count,) mean
// and M2 triples of the rows it wrote -- the statistics' first pass over C (bn_stats_stage1: one more read of C) is not launched.
//   in   ws / ws_floats   where the partials may go (the BatchNorm workspace)
//   out  done             1 = partials written: nparts of them, at ws_mean[p * n + col] / ws_m2[..] and, when chunk_rows == 0,
//                         ws_cnt[..] (per-partial row counts); chunk_rows > 0: partial p covers rows [p * chunk_rows, ..)
struct ColStats {
  float* ws; int64_t ws_floats;
  int done; int nparts; int chunk_rows; float* ws_cnt; float* ws_mean; float* ws_m2;
};
int gemm_rowpanel(const float* a, int64_t lda, int64_t m, int k, const float* w, int64_t ldw, int n, const float* ep_scale,
                  const float* ep_shift, int relu, float* c, int64_t ldc, void* stream, ColStats* cs = nullptr);
// glnn_gemm_f32 (plain A) + column statistics partials when the kernel that takes the shape can produce them (cs->done)
int gemm_stats(const float* a, int64_t lda, int64_t m, int k, const float* w, int64_t ldw, int n, const float* bias, float* c, int64_t ldc,
               float* workspace, int64_t workspace_floats, void* stream, ColStats* cs);

// mlp_lat.hip: the latency form of a student layer (m <= ~1k rows, k <= 256): C = A' * B + bias in 32-row tiles whose four waves split
// K, with the reduction that used to be the next launch as epilogue.  GLNN_ERR_UNSUPPORTED = not launched, use the tiled kernels.
struct LatStats {      // BatchNorm1d training statistics of C + finalize (arguments as bn_stats)
  const float* gamma; const float* beta; float eps; float momentum; float* running_mean; float* running_var; int64_t* nbt;
  float* mean_out; float* rstd_out; float* a_scale_out; float* a_shift_out; float* ws; int64_t ws_floats; int* counters;
};
struct LatLoss {       // log_softmax + loss + dlogits on C (n <= 64) (arguments as softmax_loss; counter and ws required)
  int kind; const int64_t* labels; const int64_t* label_rows; const float* target_logp; int64_t ldt; const int64_t* target_rows;
  float lamb; float* dlogits; int64_t ldg; float* loss_out; float* loss_accum; float* ws; int64_t ws_floats; int* counter; float* col_sum;
  struct PendingFolds* pf;      // optional: register the partials for the fused Adam launch instead of folding them here
};
int gemm_lat(const float* a, int64_t lda, const int64_t* a_rows, const float* a_scale, const float* a_shift, float drop_p,
             uint32_t drop_seed, int64_t m, int k, const float* b, int64_t ldb, int b_layout, int n, const float* bias, float* c,
             int64_t ldc, const LatStats* pend, const LatStats* st, const LatLoss* ls, void* stream, float* a_copy = nullptr,
             int64_t ld_copy = 0, const float* ep_scale = nullptr, int relu = 0);
// a_copy (plain operands only): the rows of A (gathered through a_rows) are also stored as a dense [m, k] matrix -- the first layer's
// batch rows, which the weight gradient reads again
// defer holds n + 2 entries: [n], [n + 1] = the column-sum folds of problems that carry a BatchNorm apply (nslab 0 = none)
int gemm_tn_lat(const TnProblem* problems, int n, void* stream, GradFold* defer = nullptr, float* workspace = nullptr,
                int64_t workspace_floats = 0);
int lat_dgrad_bn_bwd(const float* dz_up, int64_t ld_up, int64_t m, int k, const float* w, int64_t ldw, int n, const float* z, int64_t ldz,
                     const float* gamma, const float* mean, const float* rstd, const float* a_scale, const float* a_shift, float drop_p,
                     uint32_t drop_seed, float* da, int64_t ldda, float* dz, int64_t lddz, float* dgamma, float* dbeta, float* dz_col_sum,
                     float* workspace, int64_t workspace_floats, void* stream, GradFold* defer_colsum, int skip_apply = 0);
int bn_apply_tiles(const float* dy, int64_t lddy, const float* z, int64_t ldz, int64_t m, int n, const float* gamma, const float* mean,
                   const float* rstd, float* dz, int64_t lddz, float* dgamma, float* dbeta, float* dz_col_sum, float* workspace,
                   int64_t workspace_floats, void* stream, GradFold* defer_colsum);
int bn_finalize_tiles(const LatStats& st, int64_t m, int n, void* stream);

}  // namespace glnn
