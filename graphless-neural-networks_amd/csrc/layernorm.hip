// nn.LayerNorm(hidden) tail of a hidden layer (reference models.py:28-31, 87-90, 174-186; used by train.conf.yaml only for the
// house_class MLP, norm_type "layer"):   y = dropout(relu?(xhat * gamma + beta)),  xhat = (z - mean_row) * rstd_row,
// rstd = 1 / sqrt(var_biased + eps)  -- forward and backward.  The statistics are per ROW, so nothing here needs a cross-workgroup
// reduction except the column sums of the parameter gradients.
//   forward   one wavefront per row, the row held in registers (float4 per lane, up to 2048 columns; wider rows re-read it),
//             two-pass mean / variance (no E[x^2] - mean^2 cancellation), wave shuffle reductions;
//   backward  one 256-thread workgroup walks a chunk of kLnRows rows, thread = column quad(s): per row
//                 dy   = da * keep/(1-p) * [relu ? (xhat*gamma+beta > 0) : 1]
//                 dxh  = dy * gamma,   m1 = mean(dxh),   m2 = mean(dxh * xhat)
//                 dz   = rstd * (dxh - m1 - xhat * m2)
//             (m1, m2 through one LDS fold of a float2 per row) and keeps per-thread column partials of dgamma = sum dy*xhat,
//             dbeta = sum dy and sum dz (the bias gradient of the Linear in front); per-chunk partials -> fixed-order fold.
// The dropout mask is the library's counter-based one (glnn_common.h: drop_keep), the same the BatchNorm tail uses.
#include "glnn_common.h"

namespace {

constexpr int kLnRows = 16;          // rows per backward workgroup (32: 81 workgroups for house_class' 2580 rows)
constexpr int kMaxQuads = 8;         // float4 per lane held in registers by the forward (64 lanes x 8 x 4 = 2048 columns)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

struct LnFwdArgs {
  const float* z; int64_t ldz; int64_t rows; int h;
  const float* gamma; const float* beta; float eps; int relu;
  uint32_t thr; uint32_t seed; float dscale;
  float* y; int64_t ldy; float* mean_out; float* rstd_out;
};

__device__ __forceinline__ float4 ln_load_row4(const float* zr, int c, int h) {
  float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < h) t = *reinterpret_cast<const float4*>(zr + c);
  if (c + 1 >= h) t.y = 0.f;
  if (c + 2 >= h) t.z = 0.f;
  if (c + 3 >= h) t.w = 0.f;
  return t;
}
__device__ __forceinline__ float ln_sq4(float4 t, int c, int h, float mean) {
  const float d0 = c < h ? t.x - mean : 0.f, d1 = c + 1 < h ? t.y - mean : 0.f, d2 = c + 2 < h ? t.z - mean : 0.f,
              d3 = c + 3 < h ? t.w - mean : 0.f;
  return (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
}
__device__ __forceinline__ void ln_store4(const LnFwdArgs& a, float* yr, float4 t, int c, int64_t r, float mean, float rstd) {
  float x[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float o = 0.f;
    if (c + k < a.h) {
      o = (x[k] - mean) * rstd;
      if (a.gamma) o = fmaf(o, a.gamma[c + k], a.beta ? a.beta[c + k] : 0.f);
      if (a.relu) o = fmaxf(o, 0.f);
      if (a.thr) o = glnn::drop_keep(a.seed, a.thr, (uint32_t)r, (uint32_t)(c + k)) ? o * a.dscale : 0.f;
    }
    x[k] = o;                                      // padding columns are written as zero
  }
  *reinterpret_cast<float4*>(yr + c) = make_float4(x[0], x[1], x[2], x[3]);
}

// REGS: the row (<= 2048 columns) stays in registers between the three passes; otherwise it is re-read (L1 / L2 resident)
template <bool REGS>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const LnFwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= a.rows) return;
  const int nq = (a.h + 255) / 256;                 // float4 per lane
  const int hp = (a.h + 3) & ~3;
  const float* zr = a.z + r * a.ldz;
  float* yr = a.y + r * a.ldy;
  float4 v[kMaxQuads];
  float s = 0.f;
  if (REGS) {
#pragma unroll
    for (int j = 0; j < kMaxQuads; ++j) {
      v[j] = j < nq ? ln_load_row4(zr, (j * 64 + lane) * 4, a.h) : make_float4(0.f, 0.f, 0.f, 0.f);
      s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    }
  } else {
    for (int j = 0; j < nq; ++j) {
      const float4 t = ln_load_row4(zr, (j * 64 + lane) * 4, a.h);
      s += (t.x + t.y) + (t.z + t.w);
    }
  }
  const float mean = wave_sum(s) / (float)a.h;
  float q = 0.f;
  if (REGS) {
#pragma unroll
    for (int j = 0; j < kMaxQuads; ++j)
      if (j < nq) q += ln_sq4(v[j], (j * 64 + lane) * 4, a.h, mean);
  } else {
    for (int j = 0; j < nq; ++j) q += ln_sq4(ln_load_row4(zr, (j * 64 + lane) * 4, a.h), (j * 64 + lane) * 4, a.h, mean);
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)a.h + a.eps);
  if (lane == 0) {
    if (a.mean_out) a.mean_out[r] = mean;
    if (a.rstd_out) a.rstd_out[r] = rstd;
  }
  if (REGS) {
#pragma unroll
    for (int j = 0; j < kMaxQuads; ++j) {
      const int c = (j * 64 + lane) * 4;
      if (j < nq && c < hp) ln_store4(a, yr, v[j], c, r, mean, rstd);
    }
  } else {
    for (int j = 0; j < nq; ++j) {
      const int c = (j * 64 + lane) * 4;
      if (c < hp) ln_store4(a, yr, ln_load_row4(zr, c, a.h), c, r, mean, rstd);
    }
  }
}

struct LnBwdArgs {
  const float* da; int64_t ldda; const float* z; int64_t ldz; int64_t rows; int h;
  const float* gamma; const float* beta; const float* mean; const float* rstd; int relu;
  uint32_t thr; uint32_t seed; float dscale;
  float* dz; int64_t lddz;
  float* ws;                    // [nchunks][3][h]: per-chunk column partials of dgamma, dbeta, sum dz
  int want_param_grads; int want_dz_sum;
};

// thread t owns the column quads t, t + 256, ... (kQ of them: h <= 1024 * kQ)
template <int kQ>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const LnBwdArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * kLnRows;
  int64_t r1 = r0 + kLnRows;
  if (r1 > a.rows) r1 = a.rows;
  __shared__ float2 s_red[2][4];                    // double-buffered: one barrier per row
  float g[kQ][4], b[kQ][4], pg[kQ][4], pb[kQ][4], pz[kQ][4];
#pragma unroll
  for (int j = 0; j < kQ; ++j)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = (j * 256 + tid) * 4 + k;
      g[j][k] = (a.gamma && c < a.h) ? a.gamma[c] : 1.f;
      b[j][k] = (a.beta && c < a.h) ? a.beta[c] : 0.f;
      pg[j][k] = pb[j][k] = pz[j][k] = 0.f;
    }
  const float inv_h = 1.0f / (float)a.h;
  int buf = 0;
  // the NEXT row's operands are requested before this row's reduction and barrier (a row was: load, wait, reduce, barrier, store --
  // one memory round trip per row, 32 in a chain per workgroup); the last iteration re-requests its own row (in bounds, unused)
  float4 zn[kQ], dn[kQ];
  float mun = a.mean[r0], rsn = a.rstd[r0];
#pragma unroll
  for (int j = 0; j < kQ; ++j) {
    const int c = (j * 256 + tid) * 4;
    zn[j] = dn[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < a.h) {
      zn[j] = *reinterpret_cast<const float4*>(a.z + r0 * a.ldz + c);
      dn[j] = *reinterpret_cast<const float4*>(a.da + r0 * a.ldda + c);
    }
  }
  for (int64_t r = r0; r < r1; ++r, buf ^= 1) {
    const float mu = mun, rs = rsn;
    float4 zc[kQ], dc[kQ];
#pragma unroll
    for (int j = 0; j < kQ; ++j) { zc[j] = zn[j]; dc[j] = dn[j]; }
    {
      const int64_t rn = r + 1 < r1 ? r + 1 : r;
      mun = a.mean[rn]; rsn = a.rstd[rn];
#pragma unroll
      for (int j = 0; j < kQ; ++j) {
        const int c = (j * 256 + tid) * 4;
        if (c < a.h) {
          zn[j] = *reinterpret_cast<const float4*>(a.z + rn * a.ldz + c);
          dn[j] = *reinterpret_cast<const float4*>(a.da + rn * a.ldda + c);
        }
      }
    }
    float xh[kQ][4], dxh[kQ][4], dy[kQ][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < kQ; ++j) {
      const int c = (j * 256 + tid) * 4;
      const float4 zz = zc[j], dd = dc[j];
      const float zv[4] = {zz.x, zz.y, zz.z, zz.w}, dv[4] = {dd.x, dd.y, dd.z, dd.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float x = 0.f, d = 0.f;
        if (c + k < a.h) {
          x = (zv[k] - mu) * rs;
          d = dv[k];
          if (a.thr) d = glnn::drop_keep(a.seed, a.thr, (uint32_t)r, (uint32_t)(c + k)) ? d * a.dscale : 0.f;
          if (a.relu && !(fmaf(x, g[j][k], b[j][k]) > 0.f)) d = 0.f;
        }
        xh[j][k] = x;
        dy[j][k] = d;
        dxh[j][k] = d * g[j][k];
        s1 += dxh[j][k];
        s2 = fmaf(dxh[j][k], x, s2);
      }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) s_red[buf][wave] = make_float2(s1, s2);
    __syncthreads();
    const float2 p0 = s_red[buf][0], p1 = s_red[buf][1], p2 = s_red[buf][2], p3 = s_red[buf][3];
    const float m1 = ((p0.x + p1.x) + (p2.x + p3.x)) * inv_h, m2 = ((p0.y + p1.y) + (p2.y + p3.y)) * inv_h;
#pragma unroll
    for (int j = 0; j < kQ; ++j) {
      const int c = (j * 256 + tid) * 4;
      if (c >= ((a.h + 3) & ~3)) continue;
      float o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        o[k] = (c + k < a.h) ? rs * (dxh[j][k] - m1 - xh[j][k] * m2) : 0.f;
        pg[j][k] = fmaf(dy[j][k], xh[j][k], pg[j][k]);
        pb[j][k] += dy[j][k];
        pz[j][k] += o[k];
      }
      *reinterpret_cast<float4*>(a.dz + r * a.lddz + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
  if (a.ws) {
    float* w = a.ws + (int64_t)blockIdx.x * 3 * a.h;
#pragma unroll
    for (int j = 0; j < kQ; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = (j * 256 + tid) * 4 + k;
        if (c < a.h) {
          if (a.want_param_grads) { w[c] = pg[j][k]; w[a.h + c] = pb[j][k]; }
          if (a.want_dz_sum) w[2 * a.h + c] = pz[j][k];
        }
      }
  }
}

// out_k[c] = sum over chunks of ws[chunk][k][c], k = 0..2 (NULL outputs skipped): four interleaved chains per column -- chunk lane j sums
// chunks j, j + 4, ... (< 4 floor(n / 4)), lane 0 then the up to three left over -- folded (s0 + s1) + (s2 + s3).  One thread per
// (column, lane), the three outputs together, eight chunks' loads in flight (a thread per column walking one output after the other
// was 3 x n / 4 dependent round trips: 21 us for 81 chunks).
__global__ __launch_bounds__(256) void ln_fold_kernel(const float* __restrict__ ws, int nchunks, int h, float* dgamma, float* dbeta, float* dzsum) {
  const int lc = threadIdx.x & 63, j = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lc;
  const int cc = c < h ? c : h - 1;
  const int n4 = nchunks & ~3;
  float s[3] = {0.f, 0.f, 0.f};
  for (int i0 = j; i0 < n4; i0 += 32) {
    float t[8][3];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + 4 * u < n4 ? i0 + 4 * u : j;
#pragma unroll
      for (int k = 0; k < 3; ++k) t[u][k] = ws[((int64_t)i * 3 + k) * h + cc];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + 4 * u < n4) { s[0] += t[u][0]; s[1] += t[u][1]; s[2] += t[u][2]; }
  }
  if (j == 0)
    for (int i = n4; i < nchunks; ++i) {
#pragma unroll
      for (int k = 0; k < 3; ++k) s[k] += ws[((int64_t)i * 3 + k) * h + cc];
    }
  __shared__ float sh[3][4][64];
#pragma unroll
  for (int k = 0; k < 3; ++k) sh[k][j][lc] = s[k];
  __syncthreads();
  if (j != 0 || c >= h) return;
  float* outs[3] = {dgamma, dbeta, dzsum};
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (outs[k]) outs[k][c] = (sh[k][0][lc] + sh[k][1][lc]) + (sh[k][2][lc] + sh[k][3][lc]);
}

}  // namespace

extern "C" int glnn_layernorm_fwd_f32(const float* z, int64_t ldz, int64_t rows, int h, const float* gamma, const float* beta, float eps,
                                      int relu, float drop_p, uint32_t drop_seed, float* y, int64_t ldy, float* mean_out,
                                      float* rstd_out, void* stream) {
  GLNN_REQUIRE(z && y, "glnn_layernorm_fwd_f32: null pointer");
  GLNN_REQUIRE(rows >= 0 && h >= 1, "glnn_layernorm_fwd_f32: bad sizes");
  const int64_t hp = (h + 3) & ~3;
  GLNN_REQUIRE(ldz >= hp && ldy >= hp && ldz % 4 == 0 && ldy % 4 == 0 && glnn::aligned16(z) && glnn::aligned16(y),
               "glnn_layernorm_fwd_f32: rows must be float4 rows (leading dimensions multiples of 4, >= round4(h), 16-byte aligned)");
  GLNN_REQUIRE(gamma || !beta, "glnn_layernorm_fwd_f32: beta without gamma");
  GLNN_REQUIRE(drop_p >= 0.f && drop_p < 1.f && eps > 0.f, "glnn_layernorm_fwd_f32: drop_p in [0,1), eps > 0");
  if (rows == 0) return GLNN_OK;
  LnFwdArgs a;
  a.z = z; a.ldz = ldz; a.rows = rows; a.h = h; a.gamma = gamma; a.beta = beta; a.eps = eps; a.relu = relu;
  a.thr = glnn::drop_threshold(drop_p); a.seed = drop_seed; a.dscale = 1.0f / (1.0f - drop_p);
  a.y = y; a.ldy = ldy; a.mean_out = mean_out; a.rstd_out = rstd_out;
  const int64_t blocks = (rows + 3) / 4;
  GLNN_REQUIRE(blocks < ((int64_t)1 << 31), "glnn_layernorm_fwd_f32: too many rows for one launch");
  if (h <= 256 * kMaxQuads) hipLaunchKernelGGL(ln_fwd_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
  else hipLaunchKernelGGL(ln_fwd_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
  return glnn::check_launch("glnn_layernorm_fwd_f32");
}

extern "C" int64_t glnn_layernorm_bwd_workspace_floats(int64_t rows, int h) {
  return ((rows + kLnRows - 1) / kLnRows) * 3 * (int64_t)h;
}

extern "C" int glnn_layernorm_bwd_f32(const float* da, int64_t ldda, const float* z, int64_t ldz, int64_t rows, int h, const float* gamma,
                                      const float* beta, const float* mean, const float* rstd, int relu, float drop_p,
                                      uint32_t drop_seed, float* dz, int64_t lddz, float* dgamma, float* dbeta, float* dz_col_sum,
                                      float* workspace, int64_t workspace_floats, void* stream) {
  GLNN_REQUIRE(da && z && dz && mean && rstd, "glnn_layernorm_bwd_f32: null pointer");
  GLNN_REQUIRE(rows >= 1 && h >= 1 && h <= 4096, "glnn_layernorm_bwd_f32: rows >= 1, 1 <= h <= 4096");
  const int64_t hp = (h + 3) & ~3;
  GLNN_REQUIRE(ldda >= hp && ldz >= hp && lddz >= hp && ldda % 4 == 0 && ldz % 4 == 0 && lddz % 4 == 0 && glnn::aligned16(da) &&
                   glnn::aligned16(z) && glnn::aligned16(dz),
               "glnn_layernorm_bwd_f32: rows must be float4 rows (leading dimensions multiples of 4, >= round4(h), 16-byte aligned)");
  GLNN_REQUIRE(gamma || !beta, "glnn_layernorm_bwd_f32: beta without gamma");
  GLNN_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "glnn_layernorm_bwd_f32: dgamma and dbeta go together");
  GLNN_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "glnn_layernorm_bwd_f32: drop_p must be in [0,1)");
  const bool want_cols = dgamma || dz_col_sum;
  const int64_t need = want_cols ? glnn_layernorm_bwd_workspace_floats(rows, h) : 0;
  GLNN_REQUIRE(!want_cols || (workspace && workspace_floats >= need), "glnn_layernorm_bwd_f32: workspace needs >= %lld floats", (long long)need);
  LnBwdArgs a;
  a.da = da; a.ldda = ldda; a.z = z; a.ldz = ldz; a.rows = rows; a.h = h; a.gamma = gamma; a.beta = beta; a.mean = mean; a.rstd = rstd;
  a.relu = relu; a.thr = glnn::drop_threshold(drop_p); a.seed = drop_seed; a.dscale = 1.0f / (1.0f - drop_p);
  a.dz = dz; a.lddz = lddz; a.ws = want_cols ? workspace : nullptr; a.want_param_grads = dgamma ? 1 : 0; a.want_dz_sum = dz_col_sum ? 1 : 0;
  const int nchunks = (int)((rows + kLnRows - 1) / kLnRows);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (h <= 1024) hipLaunchKernelGGL(ln_bwd_kernel<1>, dim3(nchunks), dim3(256), 0, st, a);
  else if (h <= 2048) hipLaunchKernelGGL(ln_bwd_kernel<2>, dim3(nchunks), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(ln_bwd_kernel<4>, dim3(nchunks), dim3(256), 0, st, a);
  int rc = glnn::check_launch("glnn_layernorm_bwd_f32");
  if (rc != GLNN_OK || !want_cols) return rc;
  hipLaunchKernelGGL(ln_fold_kernel, dim3((h + 63) / 64), dim3(256), 0, st, workspace, nchunks, h, dgamma, dbeta, dz_col_sum);
  return glnn::check_launch("glnn_layernorm_bwd_f32(fold)");
}
