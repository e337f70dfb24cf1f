// One C call = forward + loss + backward of the MLP student step (reference train_and_eval.py:74-84): the
// same kernel sequence glnn_amd/student.py documents, issued from C++ so that small configurations are not
// bound by ~30 Python->ctypes round trips per step.  glnn_mlp_fwd_bwd_f32 leaves the gradient exchange (data parallel) and the
// fused Adam (glnn_adam_step_f32) to separate calls so that an all-reduce can sit between them; glnn_mlp_train_step_f32 (ABI 6)
// includes Adam and lets it fold whatever the backward left in partial form.  Batches of <= 1024 rows take the latency kernels
// of mlp_lat.hip wherever a layer qualifies (every such call returns GLNN_ERR_UNSUPPORTED without launching when it does not, and
// the tiled GEMM + separate reduction kernels follow).
#include <cstdlib>

#include "glnn_common.h"

#define GLNN_TRY(expr)          \
  do {                          \
    const int rc_ = (expr);     \
    if (rc_ != GLNN_OK) return rc_; \
  } while (0)

// pf != NULL (glnn_mlp_train_step_f32): the fused Adam launch follows immediately and is the only consumer of the gradients, so the
// final sums of gradient partials (split-K slabs of the weight gradients, ...) are left to it: they are registered in *pf instead of
// being folded by launches / last-workgroup tails of their own.
static int mlp_fwd_bwd_impl(const glnn_mlp_step_desc* d, const float* feats, int64_t ldx, const int64_t* idx,
                            int64_t m, int kind, const int64_t* labels, const float* target_logp, int64_t ldt,
                            const int64_t* target_rows, float lamb, const uint32_t* drop_seeds, void* stream, glnn::PendingFolds* pf) {
  GLNN_REQUIRE(d && feats, "glnn_mlp_fwd_bwd_f32: null pointer");
  const int L = d->num_layers;
  GLNN_REQUIRE(L >= 1 && L <= GLNN_MLP_MAX_LAYERS, "glnn_mlp_fwd_bwd_f32: num_layers=%d outside [1,%d]", L, GLNN_MLP_MAX_LAYERS);
  GLNN_REQUIRE(m >= 1 && m <= d->max_batch, "glnn_mlp_fwd_bwd_f32: batch %lld exceeds the buffers (%lld)", (long long)m, (long long)d->max_batch);
  const float p = d->dropout_p;
  GLNN_REQUIRE(p == 0.f || drop_seeds, "glnn_mlp_fwd_bwd_f32: dropout needs per-layer seeds");

  glnn::BnGroup grp_storage;
  const glnn::BnGroup* grp = nullptr;
  GLNN_REQUIRE(d->batchnorm >= 0 && d->batchnorm <= 2, "glnn_mlp_fwd_bwd_f32: batchnorm must be 0 (no norm), 1 (BatchNorm1d) or 2 (LayerNorm)");
  const bool layernorm = d->batchnorm == 2;          // per-row statistics: nothing to exchange when a batch is split over ranks
  if (d->world > 1 && d->batchnorm == 1) {
    GLNN_REQUIRE(d->exchange && d->sync_send && d->sync_recv && d->sync_rows && d->rank >= 0 && d->rank < d->world,
                 "glnn_mlp_fwd_bwd_f32: world=%d needs the exchange hook, sync buffers and a valid rank", d->world);
    grp_storage = {d->world, d->rank, d->exchange, d->exchange_ctx, d->sync_send, d->sync_recv, d->sync_rows};
    grp = &grp_storage;
  }

  // optional sync counters (zero on entry, left zero): two-stage reductions finish in their first launch (student.hip)
  int* cnt = (d->sync_counters && grp == nullptr) ? d->sync_counters : nullptr;
  // ---- forward ----
  const float* src = feats;
  int64_t ld_src = ldx;
  const int64_t* rows = idx;
  // the batch rows copied once: layer 0's GEMMs read a plain operand (float4 rows only; other inputs keep the gather in the operand loads)
  bool pregather = d->xb && idx && ldx % 4 == 0 && glnn::aligned16(feats) && ldx >= ((d->dims[0] + 3) & ~3);
  // small batches: no gather launch -- the first layer's latency GEMM stores the rows it gathers (gemm_lat a_copy); decided at layer 0
  bool lazy_copy = pregather && cnt && m <= 1024 && d->batchnorm != 2;
  if (pregather && !lazy_copy && m <= 1024) pregather = false;      // a small batch without that kernel: a gather launch costs more than it saves
  if (pregather && !lazy_copy) {
    GLNN_REQUIRE(d->ld_xb >= ((d->dims[0] + 3) & ~3), "glnn_mlp_fwd_bwd_f32: ld_xb too small");
    GLNN_TRY(glnn_gather_rows_f32(feats, ldx, idx, m, d->dims[0], d->xb, d->ld_xb, stream));
    src = d->xb;
    ld_src = d->ld_xb;
    rows = nullptr;
  }
  const float* a_scale = nullptr;
  const float* a_shift = nullptr;
  const glnn::Options& opt = glnn::opts();                            // switches between equal-result forms (tests, A/B): read once per process
  const bool narrow_bwd = opt.narrow_bwd != 0;                        // 0: always write the classifier's input gradient
  const int64_t narrow_min = opt.narrow_bwd_min;                      // rows x hidden width from which the recomputing form is used
  const bool narrow_wgrad = opt.narrow_wgrad != 0;                    // 0: the classifier's weight gradient stays a gemm_tn launch
  // Will the backward take the classifier's weight AND bias gradient out of the BatchNorm backward's first pass (bn_bwd_partial_wg_sk, see
  // the loop below)?  Then the loss kernel need not sum the bias gradient: it runs its 1024-workgroup form without the last-workgroup fold
  // (products MLP, B = 4096: 17 -> 7 us) and leaves the loss scalar to Adam.  The conditions mirror the backward's; should it fall back
  // after all, gemm_tn computes the column sums (fused_bias = false asks it to).
  bool expect_wg = false;
  if (L >= 2 && narrow_bwd && narrow_wgrad && grp == nullptr && d->batchnorm == 1 && d->dims[L] <= 64 && m > 1024 &&
      (int64_t)m * d->dims[L - 1] >= narrow_min && !d->act[L - 2] && !d->grad_ready && (m + 127) / 128 <= 64) {
    const int64_t chunks = (m + 127) / 128;
    const int64_t need = ((chunks * d->dims[L] * d->dims[L - 1] + 3) & ~(int64_t)3) + ((chunks * d->dims[L] + 3) & ~(int64_t)3);
    expect_wg = need <= d->ws_tn_floats && d->dims[L - 1] % 4 == 0 && d->ld_dlogits % 4 == 0 && d->ld_dlogits >= d->dims[L] &&
                glnn::aligned16(d->dlogits) && glnn::aligned16(d->w[L - 1]) && glnn::aligned16(d->ws_tn);
  }
  // with sync counters the loss kernel's last workgroup also finalises the loss and the last layer's bias gradient
  int* loss_cnt = (cnt && !expect_wg) ? cnt + GLNN_MLP_COUNTERS - 1 : nullptr;
  const bool fused_bias = cnt && !expect_wg && d->dims[L] <= 64 && d->ws_loss_floats >= 256 * 65;
  bool loss_done = false;
  int logit_slabs = 0;
  const bool pad_w0 = opt.pad_w0 != 0;                                // 0: wide unaligned first layers stay on the unaligned-W latency kernel
  const bool slab_consumers = opt.slab_consumers != 0;
  const bool defer_stats = opt.defer_stats != 0;
  glnn::LatStats pend = {}, next = {};
  bool have_pend = false;
  for (int l = 0; l < L; ++l) {
    const bool last = (l == L - 1);
    bool stats_done = false, have_next = false;
    float* out = last ? d->logits : d->z[l];
    const int64_t ldo = last ? d->ld_logits : d->ldz[l];
    const bool recompute = l > 0 && a_scale != nullptr;     // the previous layer's tail evaluated in this GEMM's operand load
    const float gp = recompute ? p : 0.f;
    const uint32_t gseed = (recompute && p > 0.f) ? drop_seeds[l - 1] : 0u;
    // W_0 over WIDE feature rows that are not float4-addressable (citeseer 3703, penn94 4814, cora 1433 features): a padded shadow in the
    // tail of ws_gemm, refreshed by one small launch, puts the product on the tiled split-K kernels -- the unaligned-W latency kernel
    // walks all of K in 64 workgroups (55 of the 114 us of a citeseer-shaped step)
    const float* w_l = d->w[l];
    int64_t ldw_l = d->dims[l];
    float* wsg = d->ws_gemm;
    int64_t wsg_floats = d->ws_gemm_floats;
    if (l == 0 && pad_w0 && d->dims[0] >= 512 && (d->dims[0] % 4 != 0 || !glnn::aligned16(d->w[0])) && d->ws_gemm) {
      const int64_t kp = (d->dims[0] + 3) & ~(int64_t)3, need = ((int64_t)d->dims[1] * kp + 3) & ~(int64_t)3;
      if (d->ws_gemm_floats >= need + 4ll * m * d->dims[1] && glnn::aligned16(d->ws_gemm)) {
        float* shadow = d->ws_gemm + ((d->ws_gemm_floats - need) & ~(int64_t)3);
        GLNN_TRY(glnn::pad_rows(d->w[0], d->dims[0], d->dims[1], d->dims[0], shadow, kp, stream));
        w_l = shadow; ldw_l = kp; wsg_floats = (d->ws_gemm_floats - need) & ~(int64_t)3;
      }
    }
    // small batches: the latency GEMM with the reduction behind it as epilogue (mlp_lat.hip) -- the BatchNorm statistics of a hidden
    // layer, log_softmax + loss + dlogits (+ the bias gradient) of the last one.  UNSUPPORTED = the tiled GEMM + separate kernels below
    int lat = GLNN_ERR_UNSUPPORTED;
    if (cnt && !layernorm) {
      const glnn::LatStats* pin = have_pend ? &pend : nullptr;         // the previous layer's statistics are still per-tile partials
      float* cp_dst = (l == 0 && lazy_copy) ? d->xb : nullptr;
      if (last && fused_bias) {
        const glnn::LatLoss ll = {kind, labels, kind == GLNN_LOSS_NLL ? target_rows : nullptr, target_logp, ldt,
                                  kind == GLNN_LOSS_KL ? target_rows : nullptr, lamb, d->dlogits, d->ld_dlogits, d->loss_out,
                                  d->loss_accum, d->ws_loss, d->ws_loss_floats, cnt + GLNN_MLP_COUNTERS - 1, d->gb[L - 1], pf};
        lat = glnn::gemm_lat(src, ld_src, rows, a_scale, a_shift, gp, gseed, m, d->dims[l], w_l, ldw_l, 0, d->dims[l + 1], d->b[l],
                             out, ldo, pin, nullptr, &ll, stream, cp_dst, d->ld_xb);
        loss_done = lat == GLNN_OK;
      } else if (!last && d->batchnorm == 1) {
        // statistics partials of layer l alternate between the halves of ws_bn: the consumer's workgroups read layer l-1's while
        // others already write layer l's.  Deferred (no counters) when the tail is recomputed by the next layer's GEMM.
        const int64_t half = d->ws_bn_floats / 2;
        next = {d->gamma[l], d->beta[l], d->bn_eps, d->bn_momentum, d->running_mean[l], d->running_var[l], d->nbt[l],
                d->mean[l], d->rstd[l], d->a_scale[l], d->a_shift[l], d->ws_bn + (l & 1) * half, half, (defer_stats && !d->act[l]) ? nullptr : cnt};
        lat = glnn::gemm_lat(src, ld_src, rows, a_scale, a_shift, gp, gseed, m, d->dims[l], w_l, ldw_l, 0, d->dims[l + 1], d->b[l],
                             out, ldo, pin, &next, nullptr, stream, cp_dst, d->ld_xb);
        stats_done = lat == GLNN_OK;
        have_next = stats_done && next.counters == nullptr;
      } else if (!last) {
        lat = glnn::gemm_lat(src, ld_src, rows, a_scale, a_shift, gp, gseed, m, d->dims[l], w_l, ldw_l, 0, d->dims[l + 1], d->b[l],
                             out, ldo, pin, nullptr, nullptr, stream, cp_dst, d->ld_xb);
      }
      if (lat != GLNN_OK && lat != GLNN_ERR_UNSUPPORTED) return lat;
    }
    if (l == 0 && lazy_copy && lat != GLNN_OK) {        // no latency GEMM at layer 0 after all: the gather stays in the operand loads
      lazy_copy = false;                                 // (a small batch: a gather launch of its own costs more than it saves)
      pregather = false;
    }
    if (lat != GLNN_OK && have_pend)       // the consumer is not a latency GEMM after all: finish the statistics with a launch of their own
      GLNN_TRY(glnn::bn_finalize_tiles(pend, m, d->dims[l], stream));
    have_pend = have_next;
    pend = next;
    // (round 6) a large batch in front of a narrow classifier: the product and -- when neither the bias gradient nor a last-workgroup fold is
    // asked of the loss kernel -- log_softmax + loss + dlogits behind it as ONE row-local launch (cls_block.hip); the choice depends on the
    // forward's shapes only, never on the backward's form
    if (lat != GLNN_OK && last && L >= 2 && !rows && !layernorm) {
      const glnn::ClsLoss cl = {kind, labels, kind == GLNN_LOSS_NLL ? target_rows : nullptr, target_logp, ldt,
                                kind == GLNN_LOSS_KL ? target_rows : nullptr, lamb, d->dlogits, d->ld_dlogits, d->loss_out, d->loss_accum,
                                d->ws_loss, d->ws_loss_floats, pf};
      int rc = GLNN_ERR_UNSUPPORTED;
      if (!loss_cnt && !fused_bias) {
        rc = glnn::cls_fwd(src, ld_src, a_scale, a_shift, gp, gseed, m, d->dims[l], w_l, ldw_l, d->dims[l + 1], d->b[l], out, ldo, &cl, stream);
        loss_done = rc == GLNN_OK;
      }
      if (rc == GLNN_ERR_UNSUPPORTED)
        rc = glnn::cls_fwd(src, ld_src, a_scale, a_shift, gp, gseed, m, d->dims[l], w_l, ldw_l, d->dims[l + 1], d->b[l], out, ldo, nullptr, stream);
      if (rc == GLNN_OK) lat = GLNN_OK;
      else if (rc != GLNN_ERR_UNSUPPORTED) return rc;
    }
    // a deep, narrow last layer (MLP3w4: 1024 -> 40) is split over K; its partial slabs are folded by the loss kernel, not by a launch
    if (lat != GLNN_OK && last && d->dims[L] <= 64 && slab_consumers) {
      const int rc = glnn::gemm_split_partials(src, ld_src, rows, a_scale, a_shift, gp, gseed, m, d->dims[l], w_l, ldw_l, 0,
                                               d->dims[l + 1], wsg, wsg_floats, &logit_slabs, stream);
      if (rc == GLNN_OK) lat = GLNN_OK;
      else if (rc != GLNN_ERR_UNSUPPORTED) return rc;
      else logit_slabs = 0;
    }
    // a deep hidden layer (MLP3w4: 1024 -> 1024 at B = 512) likewise: the statistics kernel folds the slabs, stores z, then reduces
    int z_slabs = 0;
    if (lat != GLNN_OK && !last && cnt && d->batchnorm == 1 && !layernorm && slab_consumers) {
      const int rc = glnn::gemm_split_partials(src, ld_src, rows, a_scale, a_shift, gp, gseed, m, d->dims[l], w_l, ldw_l, 0,
                                               d->dims[l + 1], wsg, wsg_floats, &z_slabs, stream);
      if (rc == GLNN_OK && z_slabs <= 8) lat = GLNN_OK;
      else if (rc == GLNN_OK) { GLNN_TRY(glnn::gemm_fold_partials(wsg, z_slabs, m, d->dims[l + 1], d->b[l], out, ldo, stream)); lat = GLNN_OK; z_slabs = 0; }
      else if (rc != GLNN_ERR_UNSUPPORTED) return rc;
      else z_slabs = 0;
    }
    // the tiled product of a BatchNorm layer with a plain operand: the kernel's epilogue leaves the first pass of the statistics
    // (glnn::gemm_stats -> cs.done; bn_stats below then launches the combine only)
    glnn::ColStats cs = {d->ws_bn, d->ws_bn_floats, 0, 0, 0, nullptr, nullptr, nullptr};
    if (lat != GLNN_OK && !last && d->batchnorm == 1 && !rows && !a_scale && opt.gemm_stats)
      GLNN_TRY(glnn::gemm_stats(src, ld_src, m, d->dims[l], w_l, ldw_l, d->dims[l + 1], d->b[l], out, ldo, wsg, wsg_floats, stream, &cs));
    else if (lat != GLNN_OK)
      GLNN_TRY(glnn_gemm_f32(src, ld_src, rows, a_scale, a_shift, gp, gseed, m,
                             d->dims[l], w_l, ldw_l, 0, d->dims[l + 1], nullptr, nullptr, d->b[l], 0, out, ldo,
                             wsg, wsg_floats, stream));
    if (!last && layernorm) {
      // LayerNorm -> ReLU -> dropout in one row-wise pass; the tail is always materialised (per-row statistics cannot ride in a
      // GEMM operand transform); mean[l] / rstd[l] hold the per-ROW statistics (max_batch floats each) for the backward
      GLNN_REQUIRE(d->act[l] && d->ld_act[l] >= ((d->dims[l + 1] + 3) & ~3), "glnn_mlp_fwd_bwd_f32: LayerNorm needs act[%d]", l);
      GLNN_TRY(glnn_layernorm_fwd_f32(out, ldo, m, d->dims[l + 1], d->gamma[l], d->beta[l], d->bn_eps, 1, p, p > 0.f ? drop_seeds[l] : 0u,
                                      d->act[l], d->ld_act[l], d->mean[l], d->rstd[l], stream));
      rows = nullptr;
      a_scale = a_shift = nullptr;
      src = d->act[l];
      ld_src = d->ld_act[l];
    } else if (!last) {
      if (d->batchnorm && !stats_done) {
        int rc = glnn::bn_stats(out, ldo, m, d->dims[l + 1], d->gamma[l], d->beta[l], d->bn_eps, d->bn_momentum,
                                d->running_mean[l], d->running_var[l], d->nbt[l], d->mean[l], d->rstd[l], d->a_scale[l],
                                d->a_shift[l], d->ws_bn, d->ws_bn_floats, stream, grp, cnt, z_slabs ? wsg : nullptr, z_slabs,
                                z_slabs ? d->b[l] : nullptr, cs.done ? &cs : nullptr);
        if (rc == GLNN_ERR_UNSUPPORTED && z_slabs) {          // not the one-launch form after all: fold with a launch, then the plain call
          GLNN_TRY(glnn::gemm_fold_partials(wsg, z_slabs, m, d->dims[l + 1], d->b[l], out, ldo, stream));
          rc = glnn::bn_stats(out, ldo, m, d->dims[l + 1], d->gamma[l], d->beta[l], d->bn_eps, d->bn_momentum,
                              d->running_mean[l], d->running_var[l], d->nbt[l], d->mean[l], d->rstd[l], d->a_scale[l],
                              d->a_shift[l], d->ws_bn, d->ws_bn_floats, stream, grp, cnt);
        }
        GLNN_TRY(rc);
      }
      rows = nullptr;
      if (d->act[l]) {       // the tail of hidden layer l materialised once: the next GEMM and the weight gradient read it plain
        GLNN_REQUIRE(d->ld_act[l] >= ((d->dims[l + 1] + 3) & ~3), "glnn_mlp_fwd_bwd_f32: ld_act[%d] too small", l);
        GLNN_TRY(glnn_act_fwd_f32(out, ldo, m, d->dims[l + 1], d->a_scale[l], d->a_shift[l], p, p > 0.f ? drop_seeds[l] : 0u,
                                  d->act[l], d->ld_act[l], stream));
        a_scale = a_shift = nullptr;
        src = d->act[l];
        ld_src = d->ld_act[l];
      } else {
        a_scale = d->a_scale[l];
        a_shift = d->a_shift[l];
        src = out;
        ld_src = ldo;
      }
    }
  }
  // ---- loss + dlogits ----
  if (!loss_done)
  GLNN_TRY(glnn::softmax_loss(d->logits, d->ld_logits, m, d->dims[L], kind, labels, kind == GLNN_LOSS_NLL ? target_rows : nullptr,
                              target_logp, ldt, kind == GLNN_LOSS_KL ? target_rows : nullptr, lamb, d->dlogits, d->ld_dlogits,
                              nullptr, 0, d->loss_out, d->loss_accum, d->ws_loss, d->ws_loss_floats, stream,
                              loss_cnt, fused_bias ? d->gb[L - 1] : nullptr,
                              logit_slabs ? d->ws_gemm : nullptr, logit_slabs, logit_slabs ? d->b[L - 1] : nullptr, pf));
  // ---- backward ----  (one stream: a two-stream form -- weight gradients on a second HIP stream -- was measured slower and lives in
  //      experiments/two_stream_backward.md since round 4)
  // Small-batch steps (sync counters on, one rank, no hooks, <= 3 layers): the weight gradients are NOT on the backward's critical
  // path, so they are collected and issued at the end as ONE gemm launch + ONE fold launch (glnn::gemm_tn_batch: same tile code, same
  // split plan, same fold order -> the same bits) instead of one gemm + one fold per layer: 4 launches fewer per 3-layer step
  // (arxiv MLP 0.127 -> 0.10x ms).  dz_l then has to outlive the loop: it alternates between d->dz and d->dz2.
  // (both batched kernels want every product inside glnn_gemm_tn_f32's 64 x 64 regime -- <= 64 tiles of 128 x 128|64 -- or they fall back
  //  to one gemm + one fold launch per layer: penn94's 4814 x 256 first layer.  Decide that here and keep the folds for Adam instead.)
  bool batch_shapes = true;
  for (int l = 0; l < L; ++l) {
    const int ti = (d->dims[l + 1] + 127) / 128, tj = d->dims[l] > 64 ? (d->dims[l] + 127) / 128 : 1;
    batch_shapes = batch_shapes && ti * tj <= 64;
  }
  const bool defer = batch_shapes && opt.batched_wgrad && cnt && grp == nullptr && !d->grad_ready && d->dz2 && d->ld_dz2 >= d->ld_dz && L <= 3 &&
                     m <= 1024 && fused_bias && !layernorm;      // (larger batches: neither batched kernel takes them -- per layer, folds left to Adam)      // (the last layer's bias gradient must come from the loss kernel: the batched launch has no column sums)
  glnn::TnProblem deferred[GLNN_MLP_MAX_LAYERS];
  int n_deferred = 0;
  int64_t tn_off = 0;
  // the FIRST hidden layer's dz has one consumer besides the bias / BatchNorm gradients: the first layer's weight gradient.  In the
  // one-call form with the latency kernels its apply pass rides in that product's operand loads (TnProblem::bn_z) -- one launch less
  struct { const float* z; int64_t ldz; const float* gamma; const float* mean; const float* rstd; float* ws; int64_t ws_floats; int64_t mt;
           float* dgamma; float* dbeta; float* colsum; float* dz_out; int64_t ld_out; } unapplied = {};
  const bool fuse_apply = opt.fuse_apply && pf && defer;
  // (large batches: the same idea -- the first layer's weight gradient applying the BatchNorm backward in its operand loads -- was
  //  measured break-even on MLP3w8 and slower on 512-wide students; experiments/gemm_tn_bn.md)
  // (round 6) the FIRST hidden layer's BatchNorm backward without its two passes over (da, z): the input-gradient product's epilogue stores dy
  // (da behind the tail's masks) and the tile column sums (glnn::gemm_bn_dy), one small launch turns them into the constants of
  // dz = alpha dy + beta z + gamma, and dz's ONE consumer -- the first layer's weight gradient -- evaluates that in its operand loads
  // (glnn::gemm_tn(..., bn): the pipelined kernel, plain operands); dz_0 is never written, its column sum (the bias gradient in front of the
  // BatchNorm, mathematically 0) is 0.  MLP3w8: partial 11.6 + apply 19.4 us and 160 MB of traffic leave the step.
  glnn::BnApplyA bn0 = {};
  bool have_bn0 = false;
  const float* dz = d->dlogits;
  int64_t ld_dz = d->ld_dlogits;
  for (int l = L - 1; l >= 0; --l) {
    if (defer) {
      glnn::TnProblem& q = deferred[n_deferred++];
      q = {};
      q.a = dz; q.lda = ld_dz; q.m = m; q.ka = d->dims[l + 1]; q.nb = d->dims[l]; q.c = d->gw[l]; q.ldc = d->dims[l];
      q.b_rows = nullptr; q.b_scale = q.b_shift = nullptr; q.drop_p = 0.f; q.drop_seed = 0u;
      if (l == 0) {
        q.b = pregather ? d->xb : feats; q.ldb = pregather ? d->ld_xb : ldx; q.b_rows = pregather ? nullptr : idx;
      } else if (d->act[l - 1]) {
        q.b = d->act[l - 1]; q.ldb = d->ld_act[l - 1];
      } else {
        q.b = d->z[l - 1]; q.ldb = d->ldz[l - 1]; q.b_scale = d->a_scale[l - 1]; q.b_shift = d->a_shift[l - 1];
        q.drop_p = p; q.drop_seed = p > 0.f ? drop_seeds[l - 1] : 0u;
      }
    }
    if (defer && unapplied.z) {            // this layer's dz exists only as (dy, tile partials): its weight gradient applies it in the loads
      glnn::TnProblem& q = deferred[n_deferred - 1];
      q.a = d->da; q.lda = d->ld_da;
      q.bn_z = unapplied.z; q.bn_ldz = unapplied.ldz; q.bn_gamma = unapplied.gamma; q.bn_mean = unapplied.mean; q.bn_rstd = unapplied.rstd;
      q.bn_p1 = unapplied.ws; q.bn_p2 = unapplied.ws + unapplied.mt * q.ka; q.bn_nparts = (int)unapplied.mt;
      q.bn_dgamma = unapplied.dgamma; q.bn_dbeta = unapplied.dbeta; q.bn_colsum = unapplied.colsum;
    }
    if (l == 0) {
      if (defer) break;
      // the first layer's weight gradient ends the critical path: it stays on `stream` (the aux stream is busy with the wide
      // layers' gradients) with its own workspace -- ws_gemm is idle during the backward
      // (earlier layers' slabs wait in ws_tn for Adam; if what is left cannot hold a few slabs of this product it would run unsplit --
      //  4 workgroups for vk_class' 512 x 100 over 6754 rows -- so it takes the idle ws_gemm and folds at once instead)
      const bool cramped = pf && tn_off > 0 && d->ws_tn_floats - tn_off < 8ll * d->dims[1] * d->dims[0] &&
                           d->ws_gemm_floats > d->ws_tn_floats - tn_off;
      const bool fold_later = pf && !cramped && tn_off < d->ws_tn_floats;
      float* ws0 = cramped ? d->ws_gemm : d->ws_tn + (fold_later ? tn_off : 0);
      const int64_t ws0_floats = cramped ? d->ws_gemm_floats : d->ws_tn_floats - (fold_later ? tn_off : 0);
      glnn::GradFold fw = {}, fc = {};
      int64_t used = 0;
      const int rc0 = glnn::gemm_tn(dz, ld_dz, m, d->dims[1], pregather ? d->xb : feats, pregather ? d->ld_xb : ldx, pregather ? nullptr : idx,
                            nullptr, nullptr, 0.f, 0u, d->dims[0], d->gw[0],
                            d->dims[0], (L == 1 && !fused_bias) ? d->gb[0] : nullptr, ws0, ws0_floats, stream,
                            fold_later ? &fw : nullptr, fold_later ? &fc : nullptr, &used, d->ws_tn_floats, have_bn0 ? &bn0 : nullptr);
      GLNN_TRY(rc0);
      if (fold_later) {
        GLNN_REQUIRE(pf->n + 2 <= glnn::kMaxGradFolds, "glnn_mlp_train_step_f32: too many pending gradient folds");   // (gemm_tn skipped its own fold launch)
        if (fw.nslab > 0) pf->e[pf->n++] = fw;
        if (fc.nslab > 0) pf->e[pf->n++] = fc;
        if (fw.nslab > 0 || fc.nslab > 0) tn_off += (used + 3) & ~(int64_t)3;
      }
      if (d->grad_ready) GLNN_REQUIRE(d->grad_ready(d->grad_ready_ctx, 0, stream) == 0, "glnn_mlp_fwd_bwd_f32: grad_ready hook failed (layer 0)");
      break;
    }
    const uint32_t seed = p > 0.f ? drop_seeds[l - 1] : 0u;
    // before the fused Adam launch (pf) a split reduction keeps its slabs -- and the column sums behind the last layer's bias gradient
    // their first-stage partials -- for Adam to fold: every layer then gets its own part of ws_tn (tn_off)
    auto weight_gradient = [&]() -> int {
      const bool cramped = pf && tn_off > 0 && d->ws_tn_floats - tn_off < 8ll * d->dims[l + 1] * d->dims[l] &&
                           d->ws_gemm_floats > d->ws_tn_floats - tn_off;      // see the first layer's product above
      const bool fold_later = pf && !cramped && tn_off < d->ws_tn_floats;
      glnn::GradFold fw = {}, fc = {};
      int64_t used = 0;
      float* wsp = cramped ? d->ws_gemm : d->ws_tn + (fold_later ? tn_off : 0);
      const int64_t wsf = cramped ? d->ws_gemm_floats : d->ws_tn_floats - (fold_later ? tn_off : 0);
      float* colsum = (l == L - 1 && !fused_bias) ? d->gb[l] : nullptr;
      int rc;
      if (d->act[l - 1])
        rc = glnn::gemm_tn(dz, ld_dz, m, d->dims[l + 1], d->act[l - 1], d->ld_act[l - 1], nullptr, nullptr, nullptr, 0.f, 0u,
                           d->dims[l], d->gw[l], d->dims[l], colsum, wsp, wsf, stream, fold_later ? &fw : nullptr, fold_later ? &fc : nullptr, &used, d->ws_tn_floats);
      else
        rc = glnn::gemm_tn(dz, ld_dz, m, d->dims[l + 1], d->z[l - 1], d->ldz[l - 1], nullptr, d->a_scale[l - 1], d->a_shift[l - 1],
                           p, seed, d->dims[l], d->gw[l], d->dims[l], colsum, wsp, wsf, stream, fold_later ? &fw : nullptr,
                           fold_later ? &fc : nullptr, &used, d->ws_tn_floats);   // hidden layers get their bias gradient from glnn_bn_relu_bwd_f32 below
      if (rc == GLNN_OK && fold_later) {
        GLNN_REQUIRE(pf->n + 2 <= glnn::kMaxGradFolds, "glnn_mlp_train_step_f32: too many pending gradient folds");   // (gemm_tn skipped its own fold launch)
        if (fw.nslab > 0) pf->e[pf->n++] = fw;
        if (fc.nslab > 0) pf->e[pf->n++] = fc;
        if (fw.nslab > 0 || fc.nslab > 0) tn_off += (used + 3) & ~(int64_t)3;
      }
      return rc;
    };
    int da_slabs = 0;
    auto input_gradient = [&]() -> int {
      // ws_gemm is idle during the backward: deep, skinny input gradients (B = 512, 1024 wide: 128 tiles x 32 dependent k-tiles) may
      // split their reduction
      if (cnt) {
        const int rc = glnn::gemm_lat(dz, ld_dz, nullptr, nullptr, nullptr, 0.f, 0u, m, d->dims[l + 1], d->w[l], d->dims[l], 1, d->dims[l],
                                      nullptr, d->da, d->ld_da, nullptr, nullptr, nullptr, stream);
        if (rc != GLNN_ERR_UNSUPPORTED) return rc;
        // deep input gradients (MLP3w4: K = 1024): split-K partials that the one-launch BatchNorm backward sums itself
        if (d->batchnorm == 1 && !layernorm && slab_consumers) {
          const int rs = glnn::gemm_split_partials(dz, ld_dz, nullptr, nullptr, nullptr, 0.f, 0u, m, d->dims[l + 1], d->w[l], d->dims[l], 1,
                                                   d->dims[l], d->ws_gemm, d->ws_gemm_floats, &da_slabs, stream);
          if (rs != GLNN_ERR_UNSUPPORTED) return rs;
          da_slabs = 0;
        }
      }
      return glnn_gemm_f32(dz, ld_dz, nullptr, nullptr, nullptr, 0.f, 0u, m, d->dims[l + 1], d->w[l], d->dims[l], 1, d->dims[l],
                           nullptr, nullptr, nullptr, 0, d->da, d->ld_da, d->ws_gemm, d->ws_gemm_floats, stream);
    };
    const bool alt = defer && ((L - 1 - l) & 1);            // alternate: dz of the layer above is still needed (deferred dW)
    float* dz_out = alt ? d->dz2 : d->dz;
    const int64_t ld_out = alt ? d->ld_dz2 : d->ld_dz;
    // a NARROW layer behind (the classifier) and a large batch: its input gradient is never written -- both BatchNorm backward passes
    // recompute da = dz . W on the matrix cores (student.hip, bn_bwd_*_sk; they read dz while dz_out is written: distinct buffers only)
    const bool narrow = narrow_bwd && grp == nullptr && d->batchnorm == 1 && !layernorm && d->dims[l + 1] <= 64 &&
                        (int64_t)m * d->dims[l] >= narrow_min && dz != dz_out;
    // ... and its first pass, holding dz and act(z) on chip, also leaves the narrow layer's own weight / bias gradient as row-chunk
    // partials in ws_tn (bn_bwd_partial_wg_sk) instead of a gemm_tn launch that re-reads z: folded by Adam, or by chunk_sum launches
    const int64_t wg_chunks = (m + 127) / 128;
    const int64_t wg_need = ((wg_chunks * d->dims[l + 1] * d->dims[l] + 3) & ~(int64_t)3) + ((wg_chunks * d->dims[l + 1] + 3) & ~(int64_t)3);
    const bool wg_colsum = l == L - 1 && !fused_bias;
    bool narrow_wg = narrow && narrow_wgrad && !defer && !d->act[l - 1] && !d->grad_ready && wg_chunks <= 64 && tn_off + wg_need <= d->ws_tn_floats;
    if (!defer && !narrow_wg) GLNN_TRY(weight_gradient());
    if (d->grad_ready) GLNN_REQUIRE(d->grad_ready(d->grad_ready_ctx, l, stream) == 0, "glnn_mlp_fwd_bwd_f32: grad_ready hook failed (layer %d)", l);
    // small batches: input gradient + BatchNorm backward as two launches with no wait between workgroups (mlp_lat.hip: the column
    // partial sums come out of the GEMM's epilogue, an apply kernel folds them in its prologue); each layer has its own slice of ws_bn
    bool lat_bn = false;
    if (cnt && d->batchnorm == 1 && grp == nullptr && L >= 2) {
      const int64_t per = d->ws_bn_floats / (L - 1) / 4 * 4;
      glnn::GradFold cf = {};
      bool skip = fuse_apply && l == 1 && (pregather || !idx);              // gemm_tn_lat's conditions: plain B operands, every dim <= 256
      for (int i = 0; i <= L; ++i) skip = skip && d->dims[i] <= 256;
      const int rc = glnn::lat_dgrad_bn_bwd(dz, ld_dz, m, d->dims[l + 1], d->w[l], d->dims[l], d->dims[l], d->z[l - 1], d->ldz[l - 1],
                                            d->gamma[l - 1], d->mean[l - 1], d->rstd[l - 1], d->a_scale[l - 1], d->a_shift[l - 1], p, seed,
                                            d->da, d->ld_da, dz_out, ld_out, d->ggamma[l - 1], d->gbeta[l - 1], d->gb[l - 1],
                                            d->ws_bn + (l - 1) * per, per, stream, (pf && pf->n < glnn::kMaxGradFolds) ? &cf : nullptr, skip ? 1 : 0);
      if (rc == GLNN_OK) {
        lat_bn = true;
        if (narrow_wg) { GLNN_TRY(weight_gradient()); narrow_wg = false; }     // (dz is still intact: lat_dgrad_bn_bwd wrote dz_out != dz)
        if (skip)
          unapplied = {d->z[l - 1], d->ldz[l - 1], d->gamma[l - 1], d->mean[l - 1], d->rstd[l - 1], d->ws_bn + (l - 1) * per, per, (m + 31) / 32,
                       d->ggamma[l - 1], d->gbeta[l - 1], d->gb[l - 1], dz_out, ld_out};
        if (cf.nslab > 0) pf->e[pf->n++] = cf;
      } else if (rc != GLNN_ERR_UNSUPPORTED) {
        return rc;
      }
    }
    if (lat_bn) {
      dz = dz_out;
      ld_dz = ld_out;
      continue;
    }
    float* wg_dw = narrow_wg ? d->ws_tn + tn_off : nullptr;
    float* wg_db = (narrow_wg && wg_colsum) ? wg_dw + ((wg_chunks * d->dims[l + 1] * d->dims[l] + 3) & ~(int64_t)3) : nullptr;
    const glnn::NarrowProduct np = {dz, ld_dz, d->dims[l + 1], d->w[l], d->dims[l], wg_dw, wg_db};
    if (l == 1 && !narrow && !defer && opt.bn0_in_gemm && d->batchnorm == 1 && grp == nullptr && m > 1024 && d->dims[1] % 4 == 0 &&
        (pregather || !idx) && d->da && dz != d->da) {      // (dy goes where da would have gone)
      const float* b0 = pregather ? d->xb : feats;
      const int64_t ldb0 = pregather ? d->ld_xb : ldx;
      const int64_t nparts = (m + 127) / 128;
      const int64_t wsb_floats = (pf && L >= 2) ? d->ws_bn_floats / (L - 1) / 4 * 4 : d->ws_bn_floats;      // (this layer's slice is the first)
      const int64_t ld_dy = d->ld_da;
      const int64_t ldmax = ld_dy > ldb0 ? (ld_dy > d->ldz[0] ? ld_dy : d->ldz[0]) : (ldb0 > d->ldz[0] ? ldb0 : d->ldz[0]);
      // every condition of gemm_tn(..., bn), the workspace-dependent split included (rows_per_split <= m): decided BEFORE dz is left unwritten
      const bool tn_ok = glnn::gemm_tn_takes_bn(d->da, ld_dy, m, d->dims[1], b0, ldb0, d->dims[0], d->z[0], d->ldz[0]) && m * ldmax < (1 << 28) &&
                         !(L == 1 && !fused_bias);
      const int64_t h4 = (2 * nparts * d->dims[1] + 3) & ~(int64_t)3;
      if (tn_ok && glnn::aligned16(d->ws_bn) && wsb_floats >= h4 + 3ll * d->dims[1]) {
        float* s1 = d->ws_bn;
        float* s2 = d->ws_bn + nparts * d->dims[1];
        const glnn::BnTail tail = {d->z[0], d->ldz[0], d->mean[0], d->rstd[0], d->a_scale[0], d->a_shift[0], p, seed, 1};
        const int rc = glnn::gemm_bn_dy(dz, ld_dz, m, d->dims[2], d->w[1], d->dims[1], d->dims[1], tail, d->da, ld_dy, s1, s2, stream);
        if (rc == GLNN_OK) {
          if (opt.bn0_consts_in_gemm && glnn::aligned16(d->gamma[0]) && glnn::aligned16(d->mean[0]) && glnn::aligned16(d->rstd[0]) &&
              glnn::aligned16(d->ggamma[0]) && glnn::aligned16(d->gbeta[0]) && glnn::aligned16(d->gb[0])) {
            // ... and the constants are made in the weight gradient's own prologue (every workgroup folds the 32 tile partials of its 128 columns)
            bn0 = {};
            bn0.z = d->z[0]; bn0.ldz = d->ldz[0]; bn0.p1 = s1; bn0.p2 = s2; bn0.nparts = (int)nparts; bn0.rows = m; bn0.bn_gamma = d->gamma[0];
            bn0.bn_mean = d->mean[0]; bn0.bn_rstd = d->rstd[0]; bn0.dgamma = d->ggamma[0]; bn0.dbeta = d->gbeta[0]; bn0.colsum = d->gb[0];
          } else
          GLNN_TRY(glnn::bn_bwd_parts_finish(s1, s2, (int)nparts, d->dims[1], m, d->z[0], d->ldz[0], d->gamma[0], d->mean[0], d->rstd[0],
                                             d->ws_bn + h4, d->ggamma[0], d->gbeta[0], d->gb[0], &bn0, stream));
          have_bn0 = true;
          dz = d->da;
          ld_dz = ld_dy;
          continue;
        } else if (rc != GLNN_ERR_UNSUPPORTED) {
          return rc;
        }
      }
    }
    if (!narrow) GLNN_TRY(input_gradient());
    if (layernorm) {
      GLNN_TRY(glnn_layernorm_bwd_f32(d->da, d->ld_da, d->z[l - 1], d->ldz[l - 1], m, d->dims[l], d->gamma[l - 1], d->beta[l - 1],
                                      d->mean[l - 1], d->rstd[l - 1], 1, p, seed, dz_out, ld_out, d->ggamma[l - 1], d->gbeta[l - 1],
                                      d->gb[l - 1], d->ws_bn, d->ws_bn_floats, stream));
    } else if (d->batchnorm) {
      int rc = GLNN_ERR_UNSUPPORTED;
      // deferred column sums stay in the workspace until Adam: every layer gets its own slice of ws_bn then
      glnn::GradFold cf = {};
      float* wsb = d->ws_bn;
      int64_t wsb_floats = d->ws_bn_floats;
      const int64_t need_l = 3 * ((m + 127) / 128) * d->dims[l];
      const bool dc = pf && grp == nullptr && L >= 2 && (d->ws_bn_floats / (L - 1) / 4 * 4) >= need_l && pf->n < glnn::kMaxGradFolds;
      if (dc) { wsb_floats = d->ws_bn_floats / (L - 1) / 4 * 4; wsb = d->ws_bn + (l - 1) * wsb_floats; }
      glnn::GradFold* cfp = dc ? &cf : nullptr;
      if (narrow) {
        rc = glnn::bn_relu_bwd(nullptr, 0, d->z[l - 1], d->ldz[l - 1], m, d->dims[l], d->gamma[l - 1], d->mean[l - 1],
                               d->rstd[l - 1], d->a_scale[l - 1], d->a_shift[l - 1], p, seed, dz_out, ld_out, d->ggamma[l - 1],
                               d->gbeta[l - 1], d->gb[l - 1], wsb, wsb_floats, stream, nullptr, nullptr, 1, 0, cfp, &np);
        if (rc == GLNN_ERR_UNSUPPORTED) {
          if (narrow_wg) { GLNN_TRY(weight_gradient()); narrow_wg = false; }
          GLNN_TRY(input_gradient());
        } else if (rc == GLNN_OK && narrow_wg) {
          // dW_l / db_l: wg_chunks partials each, k ascending -- by the fused Adam launch (pf) or here
          const int kcls = d->dims[l + 1];
          if (pf && pf->n + 2 <= glnn::kMaxGradFolds) {
            pf->e[pf->n++] = {d->gw[l], wg_dw, (int)wg_chunks, 0, (int64_t)kcls * d->dims[l]};
            if (wg_db) pf->e[pf->n++] = {d->gb[l], wg_db, (int)wg_chunks, 0, (int64_t)kcls};
            tn_off += wg_need;
          } else {
            GLNN_TRY(glnn::chunk_sum(wg_dw, (int)wg_chunks, kcls * d->dims[l], d->gw[l], stream));
            if (wg_db) GLNN_TRY(glnn::chunk_sum(wg_db, (int)wg_chunks, kcls, d->gb[l], stream));
          }
        }
      }
      if (da_slabs > 0)
        rc = glnn::bn_relu_bwd(d->ws_gemm, d->dims[l], d->z[l - 1], d->ldz[l - 1], m, d->dims[l], d->gamma[l - 1], d->mean[l - 1],
                               d->rstd[l - 1], d->a_scale[l - 1], d->a_shift[l - 1], p, seed, dz_out, ld_out, d->ggamma[l - 1],
                               d->gbeta[l - 1], d->gb[l - 1], wsb, wsb_floats, stream, grp, cnt, 1, da_slabs, cfp);
      if (rc == GLNN_ERR_UNSUPPORTED) {
        if (da_slabs > 0) GLNN_TRY(glnn::gemm_fold_partials(d->ws_gemm, da_slabs, m, d->dims[l], nullptr, d->da, d->ld_da, stream));
        rc = glnn::bn_relu_bwd(d->da, d->ld_da, d->z[l - 1], d->ldz[l - 1], m, d->dims[l], d->gamma[l - 1], d->mean[l - 1],
                               d->rstd[l - 1], d->a_scale[l - 1], d->a_shift[l - 1], p, seed, dz_out, ld_out, d->ggamma[l - 1],
                               d->gbeta[l - 1], d->gb[l - 1], wsb, wsb_floats, stream, grp, cnt, 1, 0, cfp);
      }
      GLNN_TRY(rc);
      if (dc && cf.nslab > 0) pf->e[pf->n++] = cf;
    } else {
      GLNN_TRY(glnn::bn_relu_bwd(d->da, d->ld_da, d->z[l - 1], d->ldz[l - 1], m, d->dims[l], nullptr, nullptr, nullptr, nullptr,
                                 nullptr, p, seed, dz_out, ld_out, nullptr, nullptr, d->gb[l - 1], d->ws_bn, d->ws_bn_floats,
                                 stream, nullptr, cnt));
    }
    dz = dz_out;
    ld_dz = ld_out;
  }
  glnn::GradFold gf[GLNN_MLP_MAX_LAYERS + 2];
  bool tn_done = false;
  int rc_lat = GLNN_ERR_UNSUPPORTED;
  if (defer) rc_lat = glnn::gemm_tn_lat(deferred, n_deferred, stream, pf ? gf : nullptr, d->ws_tn, d->ws_tn_floats);
  if (rc_lat != GLNN_OK && rc_lat != GLNN_ERR_UNSUPPORTED) return rc_lat;        // a real launch error is not a reason to fall back
  if (rc_lat == GLNN_OK) {
    // every weight gradient of the step from one launch of the latency kernel; with Adam next, its reduction slabs are folded there
    tn_done = true;
    if (pf)
      for (int i = 0; i < n_deferred + 2 && pf->n < glnn::kMaxGradFolds; ++i)
        if (i < n_deferred || gf[i].nslab > 0) pf->e[pf->n++] = gf[i];
  }
  if (defer && !tn_done && unapplied.z) {      // the latency kernel did not take the batch after all: apply with a launch, plain operand
    glnn::GradFold cf = {};
    GLNN_TRY(glnn::bn_apply_tiles(d->da, d->ld_da, unapplied.z, unapplied.ldz, m, d->dims[1], unapplied.gamma, unapplied.mean, unapplied.rstd,
                                  unapplied.dz_out, unapplied.ld_out, unapplied.dgamma, unapplied.dbeta, unapplied.colsum, unapplied.ws,
                                  unapplied.ws_floats, stream, (pf && pf->n < glnn::kMaxGradFolds) ? &cf : nullptr));
    if (cf.nslab > 0) pf->e[pf->n++] = cf;
    glnn::TnProblem& q = deferred[n_deferred - 1];
    q.a = unapplied.dz_out; q.lda = unapplied.ld_out;
    q.bn_z = nullptr;
  }
  if (tn_done) {
  } else if (defer) {
    const int rc = glnn::gemm_tn_batch(deferred, n_deferred, d->ws_tn, d->ws_tn_floats, stream, pf ? gf : nullptr);
    if (rc == GLNN_OK && pf)
      for (int i = 0; i < n_deferred && pf->n < glnn::kMaxGradFolds; ++i) pf->e[pf->n++] = gf[i];
    if (rc == GLNN_ERR_UNSUPPORTED) {          // a shape outside the 64 x 64 path (or too little workspace): one by one, as before
      for (int i = 0; i < n_deferred; ++i) {
        const glnn::TnProblem& q = deferred[i];
        GLNN_TRY(glnn_gemm_tn_f32(q.a, q.lda, q.m, q.ka, q.b, q.ldb, q.b_rows, q.b_scale, q.b_shift, q.drop_p, q.drop_seed, q.nb, q.c, q.ldc,
                                  nullptr, d->ws_tn, d->ws_tn_floats, stream));    // each with the whole workspace: the plan of the two-call form
      }
    } else if (rc != GLNN_OK) {
      return rc;
    }
  }
  return GLNN_OK;
}

extern "C" int glnn_mlp_fwd_bwd_f32(const glnn_mlp_step_desc* d, const float* feats, int64_t ldx, const int64_t* idx,
                                    int64_t m, int kind, const int64_t* labels, const float* target_logp, int64_t ldt,
                                    const int64_t* target_rows, float lamb, const uint32_t* drop_seeds, void* stream) {
  return mlp_fwd_bwd_impl(d, feats, ldx, idx, m, kind, labels, target_logp, ldt, target_rows, lamb, drop_seeds, stream, nullptr);
}

// The whole optimisation step -- forward + loss + backward + Adam -- in ONE call (reference train_and_eval.py:74-85 incl.
// optimizer.step()): glnn_mlp_fwd_bwd_f32 followed by glnn_adam_step_f32 on the same stream, for hosts that have nothing to put
// between them (no gradient exchange).  Knowing that Adam is the next launch, the backward leaves the last sums of its gradient
// partials to it (see mlp_fwd_bwd_impl).
extern "C" int glnn_mlp_train_step_f32(const glnn_mlp_step_desc* d, const float* feats, int64_t ldx, const int64_t* idx,
                                       int64_t m, int kind, const int64_t* labels, const float* target_logp, int64_t ldt,
                                       const int64_t* target_rows, float lamb, const uint32_t* drop_seeds, const glnn_adam_desc* adam,
                                       void* stream) {
  GLNN_REQUIRE(adam && adam->params && adam->grads && adam->exp_avg && adam->exp_avg_sq && adam->sizes && adam->grads_host,
               "glnn_mlp_train_step_f32: the Adam descriptor is incomplete");
  glnn::PendingFolds pf = {};
  const bool folds = glnn::opts().adam_folds && adam->num_tensors <= 32;
  GLNN_TRY(mlp_fwd_bwd_impl(d, feats, ldx, idx, m, kind, labels, target_logp, ldt, target_rows, lamb, drop_seeds, stream, folds ? &pf : nullptr));
  return glnn::adam_step(adam->params, adam->grads, adam->exp_avg, adam->exp_avg_sq, adam->sizes, adam->num_tensors, adam->max_size,
                         adam->lr, adam->beta1, adam->beta2, adam->eps, adam->weight_decay, adam->step, adam->grads_host,
                         folds ? &pf : nullptr, stream);
}
