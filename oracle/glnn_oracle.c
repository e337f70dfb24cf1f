/*
 * glnn_oracle.c -- CPU restatement of the GLNN teacher-forward arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under graphless-neural-networks_amd/ may
 * import, link or call this file.  It is used by tests/, by
 * __graft_entry__.smoke() and by bench.py's `cpu_baseline` leg as the checker
 * and as the timed CPU "port" -- never as the product path.
 *
 * PARITY STATUS: the graph arithmetic of the reference lives in the
 * un-vendored third-party package dgl==0.6.1 (reference requirements.txt:15),
 * which is absent from /root/reference and from this image, and the reference
 * holds no tests or golden vectors at that boundary ("parity unpinned" by the
 * reference itself -- SURVEY.md section 8c).  This restatement follows the
 * published semantics of the three DGL entry points the reference calls and
 * is pinned in tests/test_oracle_teacher.py by (1) hand-derived known answers
 * on tiny graphs, (2) scipy.sparse CSR matmul and (3) torch.sparse_csr matmul.
 *
 * Reference call sites restated here:
 *   - SAGEConv(in,out,"gcn")(block,(h,h_dst))   models.py:84-99, :112, :138
 *       neigh = sum_{(u->v)} h_src[u]                (update_all(copy_src,sum))
 *       h     = (neigh + h_dst[v]) / (in_deg(v) + 1)
 *       out   = h @ W_neigh^T + b                    (fc_neigh, no fc_self)
 *   - GraphConv(in,out,activation)(g,h)         models.py:170-187, :193
 *       norm='both':  h' = h * outdeg.clamp(1)^-1/2 ; agg = A h' (W before or
 *       after, whichever is cheaper) ; rst = agg * indeg.clamp(1)^-1/2 + b
 *   - g.update_all(fn.copy_u, fn.sum)           utils.py:185 (feature_prop)
 *   - per-layer BN(eval)/ReLU of SAGE.inference  models.py:139-143
 *
 * Graph layout: CSR over DESTINATION rows (row v lists the sources u of the
 * in-edges u->v), int64 indptr[n_dst+1], int32 indices[nnz]; multi-edges are
 * kept and count multiply; features row-major fp32 with a leading dimension.
 * Accumulation is plain fp32 in edge order (what a scalar CPU SpMM does).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

ORACLE_API int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* out[v,:] = row_scale[v] * sum_{e in row v} col_scale[idx[e]] * x[idx[e],:]
 * row_scale / col_scale may be NULL (treated as 1).  Plain copy_u+sum when both
 * are NULL (utils.py:185).  */
ORACLE_API void oracle_spmm_sum_f32(const int64_t* indptr, const int32_t* indices,
                                    int64_t n_dst, const float* x, int64_t ldx, int d,
                                    const float* row_scale, const float* col_scale,
                                    float* out, int64_t ldo, int threads) {
  (void)threads;
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads > 0 ? threads : 1)
  for (int64_t v = 0; v < n_dst; ++v) {
    float* o = out + v * ldo;
    for (int j = 0; j < d; ++j) o[j] = 0.0f;
    for (int64_t e = indptr[v]; e < indptr[v + 1]; ++e) {
      const int64_t u = indices[e];
      const float* xr = x + u * ldx;
      if (col_scale) {
        const float cs = col_scale[u];
        for (int j = 0; j < d; ++j) o[j] += xr[j] * cs;
      } else {
        for (int j = 0; j < d; ++j) o[j] += xr[j];
      }
    }
    if (row_scale) {
      const float rs = row_scale[v];
      for (int j = 0; j < d; ++j) o[j] *= rs;
    }
  }
}

/* SAGEConv "gcn" aggregator, before fc_neigh:
 *   out[v,:] = (sum_{u->v} x[u,:] + x_self[v,:]) / (in_deg(v) + 1)
 * x_self is h_dst = the first n_dst rows of the block's source features
 * (models.py:109,137).  Zero-in-degree rows give x_self[v]/1.  */
ORACLE_API void oracle_sage_gcn_agg_f32(const int64_t* indptr, const int32_t* indices,
                                        int64_t n_dst, const float* x, int64_t ldx, int d,
                                        const float* x_self, int64_t lds, float* out,
                                        int64_t ldo, int threads) {
  (void)threads;
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads > 0 ? threads : 1)
  for (int64_t v = 0; v < n_dst; ++v) {
    float* o = out + v * ldo;
    for (int j = 0; j < d; ++j) o[j] = 0.0f;
    for (int64_t e = indptr[v]; e < indptr[v + 1]; ++e) {
      const float* xr = x + (int64_t)indices[e] * ldx;
      for (int j = 0; j < d; ++j) o[j] += xr[j];
    }
    const float degp1 = (float)(indptr[v + 1] - indptr[v]) + 1.0f;
    const float* s = x_self + v * lds;
    for (int j = 0; j < d; ++j) o[j] = (o[j] + s[j]) / degp1;
  }
}

/* y = x @ W^T + b with W [n_out, k] row-major (torch.nn.Linear / fc_neigh
 * layout), or y = x @ W + b with W [k, n_out] (dgl GraphConv layout) when
 * w_is_in_by_out != 0.  b may be NULL.  fp32 accumulation in k order. */
ORACLE_API void oracle_linear_f32(const float* x, int64_t ldx, int64_t m, int k, const float* w,
                                  int64_t ldw, int n_out, int w_is_in_by_out, const float* b,
                                  float* y, int64_t ldy, int threads) {
  (void)threads;
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
  for (int64_t i = 0; i < m; ++i) {
    const float* xr = x + i * ldx;
    float* yr = y + i * ldy;
    for (int j = 0; j < n_out; ++j) {
      float acc = 0.0f;
      if (w_is_in_by_out) {
        for (int p = 0; p < k; ++p) acc += xr[p] * w[(int64_t)p * ldw + j];
      } else {
        const float* wr = w + (int64_t)j * ldw;
        for (int p = 0; p < k; ++p) acc += xr[p] * wr[p];
      }
      yr[j] = acc + (b ? b[j] : 0.0f);
    }
  }
}

/* nn.BatchNorm1d in eval mode (running stats), then optional ReLU; in place.
 *   y = (x - mean) / sqrt(var + eps) * gamma + beta          models.py:139-143
 * Any of mean/var/gamma/beta NULL => BN skipped entirely (norm_type "none"). */
ORACLE_API void oracle_bn_eval_relu_f32(float* x, int64_t ldx, int64_t m, int d, const float* mean,
                                        const float* var, const float* gamma, const float* beta,
                                        float eps, int relu, int threads) {
  (void)threads;
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
  for (int64_t i = 0; i < m; ++i) {
    float* r = x + i * ldx;
    for (int j = 0; j < d; ++j) {
      float v = r[j];
      if (mean && var) {
        v = (v - mean[j]) / sqrtf(var[j] + eps);
        if (gamma) v *= gamma[j];
        if (beta) v += beta[j];
      }
      if (relu && v < 0.0f) v = 0.0f;
      r[j] = v;
    }
  }
}

/* log_softmax over dim 1 (train_and_eval.py:98), in place. */
ORACLE_API void oracle_log_softmax_f32(float* x, int64_t ldx, int64_t m, int c, int threads) {
  (void)threads;
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
  for (int64_t i = 0; i < m; ++i) {
    float* r = x + i * ldx;
    float mx = r[0];
    for (int j = 1; j < c; ++j) mx = r[j] > mx ? r[j] : mx;
    double s = 0.0;
    for (int j = 0; j < c; ++j) s += exp((double)(r[j] - mx));
    const float lse = mx + (float)log(s);
    for (int j = 0; j < c; ++j) r[j] -= lse;
  }
}

/* in-degree / out-degree helpers (g.in_degrees(), g.out_degrees()). */
ORACLE_API void oracle_degrees(const int64_t* indptr, const int32_t* indices, int64_t n_dst,
                               int64_t n_src, float* in_deg, float* out_deg) {
  if (in_deg)
    for (int64_t v = 0; v < n_dst; ++v) in_deg[v] = (float)(indptr[v + 1] - indptr[v]);
  if (out_deg) {
    for (int64_t u = 0; u < n_src; ++u) out_deg[u] = 0.0f;
    for (int64_t e = 0; e < indptr[n_dst]; ++e) out_deg[indices[e]] += 1.0f;
  }
}
