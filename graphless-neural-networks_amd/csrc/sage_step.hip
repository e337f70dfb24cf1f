// One C call = forward + NLL + backward of a sampled-block GraphSAGE training step (reference train_and_eval.py:39-53:
// model(blocks, feats[input_nodes]) -> log_softmax -> NLLLoss -> (loss*lamb).backward(); SAGE.forward models.py:101-119): the
// launch sequence glnn_amd/teacher.py documents, issued from C++ so that the ~45 launches of a step cost one host round trip
// (issued from Python they took ~0.7 ms of host time per step -- more than the 0.55 ms the GPU needs on the ogbn-arxiv config).
// Every buffer is caller-owned (glnn_sage_step_desc); the optimiser step is NOT included: call glnn_adam_step_f32 next.
#include <hip/hip_runtime.h>
#include "glnn_common.h"

#define GLNN_TRY(expr)              \
  do {                              \
    const int rc_ = (expr);         \
    if (rc_ != GLNN_OK) return rc_; \
  } while (0)

// pf != NULL (glnn_sage_train_step_f32, round 6): the fused Adam launch follows immediately and is the only consumer of the gradients, so the
// last sums of gradient partials -- the split slabs of every weight gradient, the column-sum partials behind the last layer's bias gradient,
// the loss kernel's per-workgroup losses -- are registered in *pf for it instead of being folded by launches of their own (six launches of
// ~4.6 us each on the products configuration); every weight gradient then gets its own part of ws_tn (the slabs wait there for Adam).
static int sage_fwd_bwd_impl(const glnn_sage_step_desc* d, void* stream, glnn::PendingFolds* pf) {
  GLNN_REQUIRE(d && d->x && d->labels && d->dlogits && d->loss_out, "glnn_sage_fwd_bwd_f32: null pointer");
  const int L = d->num_layers;
  GLNN_REQUIRE(L >= 1 && L <= GLNN_SAGE_MAX_LAYERS, "glnn_sage_fwd_bwd_f32: num_layers=%d outside [1,%d]", L, GLNN_SAGE_MAX_LAYERS);
  const float p = d->dropout_p;
  for (int l = 0; l < L; ++l) {
    const glnn_sage_layer& y = d->layer[l];
    GLNN_REQUIRE(y.indptr && y.w && y.gw && y.agg && y.z && y.n_dst >= 1 && y.n_src >= y.n_dst,
                 "glnn_sage_fwd_bwd_f32: layer %d: null pointer or bad block sizes", l);
    GLNN_REQUIRE(l == 0 || y.n_src == d->layer[l - 1].n_dst, "glnn_sage_fwd_bwd_f32: block %d has %lld sources, block %d %lld destinations",
                 l, (long long)y.n_src, l - 1, (long long)d->layer[l - 1].n_dst);
    // (tr_ws == NULL: the caller built the transposed block and 1/(deg+1) itself -- e.g. its batch loader, off the step's stream)
    GLNN_REQUIRE(l == 0 || (y.t_indptr && y.t_indices && y.inv_deg), "glnn_sage_fwd_bwd_f32: layer %d needs the transpose buffers", l);
  }
  GLNN_REQUIRE(L == 1 || (d->dagg && d->dh), "glnn_sage_fwd_bwd_f32: backward scratch missing");

  // the transposed block + 1/(deg+1) of a layer's backward are built right in front of it (building them on a second stream under the
  // forward was measured equal and removed in round 4)
  auto transposes = [&](int l, void* st) -> int {
    const glnn_sage_layer& y = d->layer[l];
    if (!y.tr_ws) return GLNN_OK;                            // prebuilt by the caller
    GLNN_TRY(glnn_csr_transpose(y.indptr, y.indices, y.n_dst, y.n_src, y.nnz, 1, y.t_indptr, y.t_indices, y.tr_ws, y.tr_ws_bytes, st));
    return glnn_degrees_f32(y.indptr, nullptr, y.n_dst, y.n_src, 0, GLNN_DEG_INV_PLUS1, y.inv_deg, nullptr, st);
  };
  // ---- forward -----------------------------------------------------------------------------------------------------
  for (int l = 0; l < L; ++l) {
    const glnn_sage_layer& y = d->layer[l];
    const int d_in = d->dims[l], d_out = d->dims[l + 1];
    const float* src = l == 0 ? d->x : d->layer[l - 1].h;
    const int64_t ld_src = l == 0 ? d->ldx : d->layer[l - 1].ldh;
    const int64_t n_src = l == 0 ? d->x_rows : y.n_src;
    // (sum_{u->v} h[u] + h_dst[v]) / (deg + 1); the outermost block may gather from the global matrix (self_rows)
    if (l > 0 && !d->layer[l - 1].h) {
      // the hidden layer in front left only z: its tail (BatchNorm affine, ReLU, dropout) is evaluated in this layer's gather
      const glnn_sage_layer& pv = d->layer[l - 1];
      const glnn::SourceTail tail = {d->batchnorm ? pv.a_scale : nullptr, d->batchnorm ? pv.a_shift : nullptr, p, pv.drop_seed};
      GLNN_TRY(glnn::spmm_csr_tail(y.indptr, y.indices, y.n_dst, n_src, pv.z, pv.ldz, d_in, tail, y.agg, y.ld_agg, stream, y.nnz));
    } else {
      GLNN_TRY(glnn::spmm_csr_nnz(y.indptr, y.indices, y.n_dst, n_src, y.nnz, src, ld_src, d_in, GLNN_AGG_SAGE_GCN, nullptr, src, ld_src,
                                  l == 0 ? y.self_rows : nullptr, y.agg, y.ld_agg, stream));
    }
    // hidden BatchNorm layers: the projection's epilogue leaves the first pass of the column statistics (glnn::ColStats)
    glnn::ColStats cs = {d->ws_bn, d->ws_bn_floats, 0, 0, 0, nullptr, nullptr, nullptr};
    if (l < L - 1 && d->batchnorm && glnn::opts().gemm_stats)
      GLNN_TRY(glnn::gemm_stats(y.agg, y.ld_agg, y.n_dst, d_in, y.w, d_in, d_out, y.b, y.z, y.ldz, d->ws_gemm, d->ws_gemm_floats, stream, &cs));
    else
      GLNN_TRY(glnn_gemm_f32(y.agg, y.ld_agg, nullptr, nullptr, nullptr, 0.f, 0u, y.n_dst, d_in, y.w, d_in, 0, d_out, nullptr, nullptr, y.b, 0,
                             y.z, y.ldz, d->ws_gemm, d->ws_gemm_floats, stream));
    if (l == L - 1) break;
    if (d->batchnorm)
      GLNN_TRY(glnn::bn_stats(y.z, y.ldz, y.n_dst, d_out, y.gamma, y.beta, d->bn_eps, d->bn_momentum, y.running_mean, y.running_var, y.nbt,
                              y.mean, y.rstd, y.a_scale, y.a_shift, d->ws_bn, d->ws_bn_floats, stream, nullptr, nullptr, nullptr, 0, nullptr,
                              cs.done ? &cs : nullptr));
    if (y.h)          // h = tail(z) materialised (optional: with h == NULL the next layer's gather evaluates the tail itself)
      GLNN_TRY(glnn_act_fwd_f32(y.z, y.ldz, y.n_dst, d_out, d->batchnorm ? y.a_scale : nullptr, d->batchnorm ? y.a_shift : nullptr, p,
                                y.drop_seed, y.h, y.ldh, stream));
  }
  // ---- loss + dlogits (labels indexed by the batch's output nodes) ----------------------------------------------------
  const glnn_sage_layer& top = d->layer[L - 1];
  GLNN_TRY(glnn::softmax_loss(top.z, top.ldz, top.n_dst, d->dims[L], GLNN_LOSS_NLL, d->labels, d->label_rows, nullptr, 0, nullptr, d->lamb,
                              d->dlogits, d->ld_dlogits, nullptr, 0, d->loss_out, d->loss_accum, d->ws_loss, d->ws_loss_floats, stream, nullptr, nullptr,
                              nullptr, 0, nullptr, pf));
  // ---- backward --------------------------------------------------------------------------------------------------------
  const float* dz = d->dlogits;
  int64_t ld_dz = d->ld_dlogits;
  // Layer 0's dz has ONE consumer, dW_0 (the outermost block's input needs no gradient).  Its BatchNorm backward then stops after the pass
  // that leaves dy and the column sums; dz = alpha dy + beta z + gamma is evaluated on dW_0's operand pieces (glnn::gemm_tn(..., bn)) and
  // the apply pass -- read da and z, write dz: 1.5 GB on the products configuration -- does not exist (round 5)
  glnn::BnApplyA ax0 = {};
  bool apply_in_gemm = false;
  int64_t tn_off = 0;                                        // pf: the part of ws_tn whose slabs already wait for Adam
  for (int l = L - 1; l >= 0; --l) {
    const glnn_sage_layer& y = d->layer[l];
    const int d_in = d->dims[l], d_out = d->dims[l + 1];
    // dW_l = dz^T agg (+ db for the last layer; hidden layers get it from the activation backward below)
    // (pf: a product whose remaining part of ws_tn cannot hold eight slabs folds at once in the whole workspace instead -- never unsplit)
    const bool later = pf && d->ws_tn && d->ws_tn_floats - tn_off >= 8ll * d_in * d_out + 64ll * d_out && pf->n + 2 <= glnn::kMaxGradFolds;
    glnn::GradFold fw = {}, fc = {};
    int64_t used = 0;
    // (slabs of earlier products wait in [0, tn_off) for Adam: a product that cannot keep its own there still works BEHIND them, in whatever is
    //  left -- fewer splits, folded at once; glnn_amd/teacher.py sizes ws_tn for every layer's slabs so that this does not happen)
    const bool behind = later || (pf && tn_off > 0);
    float* wsp = d->ws_tn ? d->ws_tn + (behind ? tn_off : 0) : nullptr;
    const int64_t wsf = d->ws_tn_floats - (behind ? tn_off : 0);
    auto keep_folds = [&]() {
      if (!later) return;
      if (fw.nslab > 0) pf->e[pf->n++] = fw;
      if (fc.nslab > 0) pf->e[pf->n++] = fc;
      if (fw.nslab > 0 || fc.nslab > 0) tn_off += (used + 3) & ~(int64_t)3;
    };
    if (l == 0 && apply_in_gemm) {
      const int rcw = glnn::gemm_tn(dz, ld_dz, y.n_dst, d_out, y.agg, y.ld_agg, nullptr, nullptr, nullptr, 0.f, 0u, d_in, y.gw, d_in, nullptr,
                                    wsp, wsf, stream, later ? &fw : nullptr, nullptr, &used, d->ws_tn_floats, &ax0);
      if (rcw == GLNN_ERR_UNSUPPORTED)                     // (dz was not written: there is nothing to fall back to)
        return glnn::fail(GLNN_ERR_UNSUPPORTED, "glnn_sage_fwd_bwd_f32: the weight-gradient workspace (%lld floats) is too small for the "
                          "deferred BatchNorm backward of layer 0; set GLNN_SAGE_FUSE_BN_APPLY=0 or enlarge ws_tn", (long long)d->ws_tn_floats);
      GLNN_TRY(rcw);
      keep_folds();
    } else {
      GLNN_TRY(glnn::gemm_tn(dz, ld_dz, y.n_dst, d_out, y.agg, y.ld_agg, nullptr, nullptr, nullptr, 0.f, 0u, d_in, y.gw, d_in,
                             l == L - 1 ? y.gb : nullptr, wsp, wsf, stream, later ? &fw : nullptr, later ? &fc : nullptr, &used, d->ws_tn_floats));
      keep_folds();
    }
    if (l == 0) break;                                     // the outermost block's input is feats: no gradient needed
    // dagg = dz W ;  dh = (A^T + I_dst)(dagg / (deg + 1)) over the transposed block
    GLNN_TRY(glnn_gemm_f32(dz, ld_dz, nullptr, nullptr, nullptr, 0.f, 0u, y.n_dst, d_out, y.w, d_in, 1, d_in, nullptr, nullptr, nullptr, 0,
                           d->dagg, d->ld_dagg, nullptr, 0, stream));
    GLNN_TRY(transposes(l, stream));
    const glnn_sage_layer& prev = d->layer[l - 1];         // its tail produced h_l: dz_{l-1} in place on dh
    // layer 0's dz has one consumer (see above): its BatchNorm backward is deferred to dW_0's operand loads when that product's shape allows
    const bool defer = l == 1 && d->batchnorm && glnn::opts().sage_fuse_bn_apply && y.n_src == prev.n_dst &&
                       glnn::gemm_tn_takes_bn(d->dh, d->ld_dh, prev.n_dst, d_in, prev.agg, prev.ld_agg, d->dims[0], prev.z, prev.ldz, 8ll * d->dims[0] * d_in) && d->ws_tn &&      // (planned against the eight slabs that are guaranteed below)
                       (int64_t)d->dims[0] * d_in * 8 <= d->ws_tn_floats;     // (room for >= 8 split slabs: every split stays inside the descriptor window)
    int rcb = GLNN_ERR_UNSUPPORTED;
    if (defer && glnn::opts().sage_fuse_bn_dy) {
      // ... and its first pass (dy, column sums) is the epilogue of the transposed aggregation: da is never written
      const glnn::BnTail tail = {prev.z, prev.ldz, prev.mean, prev.rstd, prev.a_scale, prev.a_shift, p, prev.drop_seed, 1};
      int nslots = 0;
      rcb = glnn::spmm_csr_bn_dy(y.t_indptr, y.t_indices, y.n_src, y.n_dst, d->dagg, d->ld_dagg, d_in, y.inv_deg, tail, d->dh, d->ld_dh, d->ws_bn,
                                 d->ws_bn_floats - (5ll * d_in + 8), &nslots, stream);
      if (rcb != GLNN_OK && rcb != GLNN_ERR_UNSUPPORTED) return rcb;
      if (rcb == GLNN_OK) {
        GLNN_TRY(glnn::bn_bwd_deferred_finish(d->ws_bn, d->ws_bn_floats, nslots, d_in, prev.n_dst, prev.z, prev.ldz, prev.gamma, prev.mean, prev.rstd,
                                              prev.ggamma, prev.gbeta, prev.gb, &ax0, stream));
        apply_in_gemm = true;
      }
    }
    if (rcb == GLNN_ERR_UNSUPPORTED) {
      GLNN_TRY(glnn::spmm_csr_nnz(y.t_indptr, y.t_indices, y.n_src, y.n_dst, y.nnz + y.n_dst, d->dagg, d->ld_dagg, d_in, GLNN_AGG_SUM, y.inv_deg,
                                  nullptr, 0, nullptr, d->dh, d->ld_dh, stream));
      if (defer) {
        rcb = glnn::bn_relu_bwd(d->dh, d->ld_dh, prev.z, prev.ldz, prev.n_dst, d_in, prev.gamma, prev.mean, prev.rstd, prev.a_scale, prev.a_shift, p,
                                prev.drop_seed, d->dh, d->ld_dh, prev.ggamma, prev.gbeta, prev.gb, d->ws_bn, d->ws_bn_floats, stream, nullptr, nullptr, 1,
                                0, nullptr, nullptr, &ax0);
        if (rcb != GLNN_OK && rcb != GLNN_ERR_UNSUPPORTED) return rcb;
        apply_in_gemm = rcb == GLNN_OK;
      }
      if (rcb == GLNN_ERR_UNSUPPORTED)
        GLNN_TRY(glnn::bn_relu_bwd(d->dh, d->ld_dh, prev.z, prev.ldz, prev.n_dst, d_in, d->batchnorm ? prev.gamma : nullptr, prev.mean, prev.rstd,
                                   d->batchnorm ? prev.a_scale : nullptr, d->batchnorm ? prev.a_shift : nullptr, p, prev.drop_seed, d->dh,
                                   d->ld_dh, prev.ggamma, prev.gbeta, prev.gb, d->ws_bn, d->ws_bn_floats, stream, nullptr));
    }
    dz = d->dh;
    ld_dz = d->ld_dh;
  }
  return GLNN_OK;
}

extern "C" int glnn_sage_fwd_bwd_f32(const glnn_sage_step_desc* d, void* stream) { return sage_fwd_bwd_impl(d, stream, nullptr); }

// The whole optimisation step of the sampled-block teacher -- forward + NLL + backward + Adam (reference train_and_eval.py:39-53 incl.
// optimizer.step()) -- in ONE call (ABI 11): glnn_sage_fwd_bwd_f32 followed by glnn_adam_step_f32 on the same stream, for hosts that have
// nothing to put between them (no gradient exchange), with the backward's last partial sums left to the Adam launch.  Same partials, same
// fold order: the same parameters, moments and loss as the two calls, bit for bit.
extern "C" int glnn_sage_train_step_f32(const glnn_sage_step_desc* d, const glnn_adam_desc* adam, void* stream) {
  GLNN_REQUIRE(adam && adam->params && adam->grads && adam->exp_avg && adam->exp_avg_sq && adam->sizes && adam->grads_host,
               "glnn_sage_train_step_f32: the Adam descriptor is incomplete");
  glnn::PendingFolds pf = {};
  const bool folds = glnn::opts().adam_folds && adam->num_tensors <= 32;
  GLNN_TRY(sage_fwd_bwd_impl(d, stream, folds ? &pf : nullptr));
  return glnn::adam_step(adam->params, adam->grads, adam->exp_avg, adam->exp_avg_sq, adam->sizes, adam->num_tensors, adam->max_size, adam->lr,
                         adam->beta1, adam->beta2, adam->eps, adam->weight_decay, adam->step, adam->grads_host, folds ? &pf : nullptr, stream);
}
