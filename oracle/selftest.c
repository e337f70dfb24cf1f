/* Sanitizer self-test of the C oracle (test infrastructure; SURVEY.md section 5, "race detection / sanitizers"): every exported function of
 * glnn_oracle.c on hand-made graphs whose answers can be written down -- a path, a star, an isolated node, a duplicate edge, a self-loop, a block
 * with fewer destinations than sources -- with every buffer allocated at EXACTLY its size, so that AddressSanitizer / UBSan (make -C oracle
 * selftest_asan) flag any out-of-range index; run by tests/test_oracle_sanitizer.py.  Exit code 0 = every known answer met.
 * The graph: 5 sources, 4 destinations (dst rows first among the sources).  In-edges:
 *   v0 <- {1, 2}      v1 <- {0, 0}  (duplicate edge: counts twice, reference dataloader.py:75-76)      v2 <- {2} (self-loop)      v3 <- {} (isolated) */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int oracle_max_threads(void);
void oracle_spmm_sum_f32(const int64_t*, const int32_t*, int64_t, const float*, int64_t, int, const float*, const float*, float*, int64_t, int);
void oracle_sage_gcn_agg_f32(const int64_t*, const int32_t*, int64_t, const float*, int64_t, int, const float*, int64_t, float*, int64_t, int);
void oracle_linear_f32(const float*, int64_t, int64_t, int, const float*, int64_t, int, int, const float*, float*, int64_t, int);
void oracle_bn_eval_relu_f32(float*, int64_t, int64_t, int, const float*, const float*, const float*, const float*, float, int, int);
void oracle_log_softmax_f32(float*, int64_t, int64_t, int, int);
void oracle_degrees(const int64_t*, const int32_t*, int64_t, int64_t, float*, float*);

static int failures = 0;
static void expect(const char* what, const float* got, const float* want, int n, float tol) {
  for (int i = 0; i < n; ++i)
    if (!(fabsf(got[i] - want[i]) <= tol)) {
      fprintf(stderr, "FAIL %s[%d]: got %.9g want %.9g\n", what, i, got[i], want[i]);
      ++failures;
    }
}
static void* exact(const void* src, size_t bytes) {          /* a heap copy of exactly `bytes` bytes: the red zones start right behind it */
  void* p = malloc(bytes ? bytes : 1);
  if (src) memcpy(p, src, bytes);
  return p;
}

int main(void) {
  const int64_t indptr_h[5] = {0, 2, 4, 5, 5};
  const int32_t indices_h[5] = {1, 2, 0, 0, 2};
  const float x_h[5 * 2] = {1, 10, 2, 20, 3, 30, 4, 40, 5, 50};          /* x[u] = (u + 1, 10 (u + 1)) */
  int64_t* indptr = exact(indptr_h, sizeof indptr_h);
  int32_t* indices = exact(indices_h, sizeof indices_h);
  float* x = exact(x_h, sizeof x_h);
  const int threads = oracle_max_threads() > 4 ? 4 : oracle_max_threads();
  if (threads < 1) { fprintf(stderr, "FAIL oracle_max_threads\n"); return 1; }

  /* copy_u + sum (utils.py:185) */
  float* out = exact(NULL, 4 * 2 * sizeof(float));
  oracle_spmm_sum_f32(indptr, indices, 4, x, 2, 2, NULL, NULL, out, 2, threads);
  { const float want[8] = {5, 50, 2, 20, 3, 30, 0, 0}; expect("spmm_sum", out, want, 8, 0.f); }
  /* GraphConv norm='both' style scales: row_scale on destinations, col_scale on sources */
  { const float rs_h[4] = {0.5f, 2.f, 1.f, 3.f}, cs_h[5] = {1.f, 2.f, 0.5f, 7.f, 7.f};
    float* rs = exact(rs_h, sizeof rs_h); float* cs = exact(cs_h, sizeof cs_h);
    oracle_spmm_sum_f32(indptr, indices, 4, x, 2, 2, rs, cs, out, 2, threads);
    const float want[8] = {0.5f * (2 * 2 + 3 * 0.5f), 0.5f * (20 * 2 + 30 * 0.5f), 2.f * 2, 2.f * 20, 1.5f, 15.f, 0, 0};
    expect("spmm_sum(scaled)", out, want, 8, 1e-6f);
    free(rs); free(cs); }
  /* SAGEConv "gcn": (sum + self) / (deg + 1); the isolated row is its own features */
  oracle_sage_gcn_agg_f32(indptr, indices, 4, x, 2, 2, x, 2, out, 2, threads);
  { const float want[8] = {(5 + 1) / 3.f, (50 + 10) / 3.f, (2 + 2) / 3.f, (20 + 20) / 3.f, (3 + 3) / 2.f, (30 + 30) / 2.f, 4, 40};
    expect("sage_gcn_agg", out, want, 8, 1e-6f); }
  /* degrees: in-degree per destination, out-degree per source (duplicates counted) */
  { float* din = exact(NULL, 4 * sizeof(float)); float* dout = exact(NULL, 5 * sizeof(float));
    oracle_degrees(indptr, indices, 4, 5, din, dout);
    const float wi[4] = {2, 2, 1, 0}, wo[5] = {2, 1, 2, 0, 0};
    expect("in_degrees", din, wi, 4, 0.f); expect("out_degrees", dout, wo, 5, 0.f);
    free(din); free(dout); }
  /* Linear in both weight layouts: [n_out, k] (torch) and [k, n_out] (dgl GraphConv) */
  { const float w_nk_h[3 * 2] = {1, 0, 0, 1, 1, 1}, w_kn_h[2 * 3] = {1, 0, 1, 0, 1, 1}, b_h[3] = {0.5f, -0.5f, 0.f};
    float* w_nk = exact(w_nk_h, sizeof w_nk_h); float* w_kn = exact(w_kn_h, sizeof w_kn_h); float* b = exact(b_h, sizeof b_h);
    float* y = exact(NULL, 5 * 3 * sizeof(float));
    float want[15];
    for (int i = 0; i < 5; ++i) { want[3 * i] = (i + 1) + 0.5f; want[3 * i + 1] = 10.f * (i + 1) - 0.5f; want[3 * i + 2] = 11.f * (i + 1); }
    oracle_linear_f32(x, 2, 5, 2, w_nk, 2, 3, 0, b, y, 3, threads);
    expect("linear[n,k]", y, want, 15, 1e-5f);
    oracle_linear_f32(x, 2, 5, 2, w_kn, 3, 3, 1, b, y, 3, threads);
    expect("linear[k,n]", y, want, 15, 1e-5f);
    for (int i = 0; i < 15; ++i) want[i] -= b_h[i % 3];
    oracle_linear_f32(x, 2, 5, 2, w_nk, 2, 3, 0, NULL, y, 3, threads);
    expect("linear(no bias)", y, want, 15, 1e-5f);
    /* BatchNorm eval + ReLU in place (models.py:139-143), and the norm-free path */
    const float mean_h[3] = {3, 30, 33}, var_h[3] = {3, 300, 363}, g_h[3] = {2, 1, -1}, be_h[3] = {0, 1, 0};
    float* mean = exact(mean_h, sizeof mean_h); float* var = exact(var_h, sizeof var_h); float* g = exact(g_h, sizeof g_h); float* be = exact(be_h, sizeof be_h);
    float bnw[15];
    for (int i = 0; i < 5; ++i)
      for (int j = 0; j < 3; ++j) {
        float v = (want[3 * i + j] - mean_h[j]) / sqrtf(var_h[j] + 1e-5f) * g_h[j] + be_h[j];
        bnw[3 * i + j] = v < 0.f ? 0.f : v;
      }
    oracle_bn_eval_relu_f32(y, 3, 5, 3, mean, var, g, be, 1e-5f, 1, threads);
    expect("bn_eval_relu", y, bnw, 15, 1e-6f);
    oracle_bn_eval_relu_f32(y, 3, 5, 3, NULL, NULL, NULL, NULL, 1e-5f, 0, threads);          /* norm "none", no ReLU: unchanged */
    expect("bn_eval(none)", y, bnw, 15, 0.f);
    /* log_softmax rows sum to one in probability space, and a constant row is -log(c) */
    oracle_log_softmax_f32(y, 3, 5, 3, threads);
    for (int i = 0; i < 5; ++i) {
      const float s = expf(y[3 * i]) + expf(y[3 * i + 1]) + expf(y[3 * i + 2]), one = 1.f;
      expect("log_softmax(sum p)", &s, &one, 1, 1e-6f);
    }
    float cst[3] = {7.f, 7.f, 7.f}, wc[3] = {-logf(3.f), -logf(3.f), -logf(3.f)};
    float* c3 = exact(cst, sizeof cst);
    oracle_log_softmax_f32(c3, 3, 1, 3, 1);
    expect("log_softmax(const)", c3, wc, 3, 1e-6f);
    free(c3); free(w_nk); free(w_kn); free(b); free(y); free(mean); free(var); free(g); free(be); }
  /* empty inputs are no-ops */
  oracle_spmm_sum_f32(indptr, indices, 0, x, 2, 2, NULL, NULL, out, 2, threads);
  oracle_linear_f32(x, 2, 0, 2, x, 2, 1, 0, NULL, out, 1, threads);
  free(indptr); free(indices); free(x); free(out);
  if (failures) { fprintf(stderr, "%d known answers missed\n", failures); return 1; }
  printf("oracle selftest: all known answers met (%d threads)\n", threads);
  return 0;
}
