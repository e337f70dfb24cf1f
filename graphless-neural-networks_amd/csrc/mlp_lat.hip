// The LATENCY form of the student's dense layers (reference MLP.forward, models.py:42-53, inside the B = 512 steps of
// train_and_eval.py:74-85), gfx950.
//
// At B = 512 a layer of the arxiv student is 32 tiles of 64 x 64: the tiled kernels of gemm.hip run 8 dependent k-tiles per
// workgroup (12 us for 0.07 GFLOP), and every reduction that follows a GEMM (BatchNorm statistics, the loss) is one more launch
// of ~6 us whose only input is what the GEMM just wrote.  Here instead
//   * a workgroup owns a 32 x 32 (hidden layers) or 32 x 64 (the <= 64-wide output layer) tile of C and its FOUR WAVES SPLIT K:
//     wave q multiplies rows x columns over its quarter of the reduction straight from global memory into MFMA fragments (no LDS
//     staging, no barrier in the loop; all of a wave's loads for up to 64 reduction terms are in flight at once), so the dependent
//     MFMA chain is K/8 instead of K/2 long and there are 4x as many workgroups as 64 x 64 tiles;
//   * the four partial tiles are summed in LDS in fixed order (k ascending), bias added, the tile stored, and THEN, still in the
//     same launch, the epilogue that used to be the next kernel:
//       STATS: per-tile (mean, M2) of the 32 rows of every column -> partials; the last workgroup of a 64-column block runs the
//              Chan combine and the whole BatchNorm finalize (running statistics, a_scale / a_shift) -- student_dev.h;
//       LOSS : log_softmax + NLL | KL(log-target) and d(lamb * loss)/dlogits on the complete rows of the tile, per-workgroup loss and
//              bias-gradient partials, last workgroup folds them (same arithmetic per row as softmax_loss_kernel).
//       BNBWD: (input-gradient product of a hidden layer) C -> dy = relu'(bn(z)) * dropout'(C) before the store + the tile's column sums of
//              dy and dy * xhat; bn_apply_tiles_kernel behind it is the BatchNorm backward without any wait between workgroups.
// Results differ from the tiled kernels only by the summation order over k (four ascending quarters, then q = 0..3).
// Also here: gemm_tn_lat_kernel (all weight gradients of a small step from one launch, same loading style).
#include <cstdlib>

#include "glnn_common.h"
#include "student_dev.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { EPI_PLAIN = 0, EPI_STATS = 1, EPI_LOSS = 2, EPI_BNBWD = 3 };
constexpr int kGroups = 8;     // k-groups (8 reduction terms: 4 per lane half) whose loads a wave keeps in flight together

struct LatArgs {
  const float* a; int64_t lda; const int64_t* a_rows; const float* a_scale; const float* a_shift;
  uint32_t drop_thr, drop_seed; float drop_scale;
  int64_t m; int k; const float* b; int64_t ldb; int n;
  const float* bias; float* c; int64_t ldc; int c_vec;
  const float* ep_scale; int relu;                // plain epilogue only: C = relu?(acc * ep_scale[col] + bias[col]) (eval-mode BatchNorm folded into the layer)
  int b_vec;                                      // W[n, k] rows float4-addressable (ldb % 4 == 0, 16-byte base); else four dword loads per fragment
  float* a_copy; int64_t ld_copy;                 // optional: the (gathered) rows of A stored as a plain [m, k] matrix by the blockIdx.y == 0 tiles
  int gpw;                                        // k-groups per wave: wave q owns groups [q * gpw, (q + 1) * gpw)
  float* ws_mean; float* ws_m2; int* counters;    // EPI_STATS: [m tiles][n] partials, one counter per 64-column block
  // EPI_BNBWD (the input-gradient product C = dz_{l+1} W_{l+1} of a hidden layer with BatchNorm): C is turned into
  // dy = relu'(bn(z)) * dropout'(C) before it is stored, and the tile's column sums of dy and dy * xhat go to e_ws1 / e_ws2 [m tiles][n]
  const float* e_z; int64_t e_ldz; const float* e_mean; const float* e_rstd; const float* e_sc; const float* e_sh;
  uint32_t e_thr, e_seed; float e_dscale; int e_relu; float* e_ws1; float* e_ws2;
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 keep_first(float4 v, int left) {   // elements t < left stay, the rest read as 0
  v.x = left > 0 ? v.x : 0.f; v.y = left > 1 ? v.y : 0.f; v.z = left > 2 ? v.z : 0.f; v.w = left > 3 ? v.w : 0.f;
  return v;
}

// XF: 0 = A as stored; 1 = relu(a * scale[k] + shift[k]); 2 = 1 followed by the counter-hash dropout of glnn_common.h.
// B_KN: B given as W[k, n] (the input-gradient product) instead of W[n, k].  NB: 32-column MFMA blocks per workgroup tile.
// FIN (XF >= 1 only): a_scale / a_shift are not in memory yet -- the producing launch left per-tile (mean, M2) partials of A's
// columns (`pend`); every workgroup combines them itself while its operand loads are in flight, and workgroup (0, 0) also stores what
// the statistics kernel would have (mean / rstd / a_scale / a_shift for the backward, the running statistics).
constexpr int kFinMaxK = 1024;
// BU (first layers only: XF = 0, W as [n, k]): W's rows are not float4-addressable -- four dword loads per fragment.
template <int XF, bool B_KN, int NB, int EPI, bool FIN, bool BU = false>
__global__ __launch_bounds__(256) void gemm_lat_kernel(const LatArgs g, const BnFinArgs fin, const LossArgs ls, const BnFinArgs pend) {
  constexpr int TN = 32 * NB;
  constexpr int LDT = TN + 4;
  __shared__ __attribute__((aligned(16))) float red[4][32 * LDT];
  __shared__ __attribute__((aligned(16))) float s_sc[FIN ? kFinMaxK : 4], s_sh[FIN ? kFinMaxK : 4];
  const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6, li = lane & 31, kk = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * 32;
  const int n0 = blockIdx.y * TN;

  // LOSS epilogue mapping: thread = (row tid >> 3 of the tile, classes 8 (tid & 7) .. + 7).  Its targets are requested before the
  // product so that they have arrived when the rows are complete.
  float tv[8];
  int64_t yv = 0;
  if (EPI == EPI_LOSS) {
    int64_t row = m0 + (tid >> 3);
    if (row > g.m - 1) row = g.m - 1;
    if (ls.kind == GLNN_LOSS_KL) {
      const float* tr = ls.t + (ls.t_rows ? ls.t_rows[row] : row) * ls.ldt;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int cls = 8 * (tid & 7) + t;
        tv[t] = tr[cls < ls.c ? cls : ls.c - 1];
      }
    } else {
      yv = ls.labels[ls.label_rows ? ls.label_rows[row] : row];
    }
  }

  // BNBWD epilogue mapping = the output mapping below (row tid >> 3, columns (tid & 7) * 4 ..): its z values and column constants are
  // requested before the product
  float4 ez = make_float4(0.f, 0.f, 0.f, 0.f), emu = ez, ers = ez, esc = ez, esh = ez;
  if (EPI == EPI_BNBWD) {
    int64_t er = m0 + (tid >> 3);
    if (er > g.m - 1) er = g.m - 1;
    int ec = n0 + (tid & 7) * 4;
    if (ec > g.n - 4) ec = g.n - 4;                    // n % 4 == 0 (host): a clamped float4 stays inside the row
    ez = ld4(g.e_z + er * g.e_ldz + ec);
    emu = ld4(g.e_mean + ec); ers = ld4(g.e_rstd + ec); esc = ld4(g.e_sc + ec); esh = ld4(g.e_sh + ec);
  }

  int64_t mrow = m0 + li;
  if (mrow > g.m - 1) mrow = g.m - 1;
  const float* ap = g.a + (g.a_rows ? g.a_rows[mrow] : mrow) * g.lda;
  const float* bp[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    int ng = n0 + 32 * j + li;
    if (ng > g.n - 1) ng = g.n - 1;
    bp[j] = B_KN ? g.b + ng : g.b + (int64_t)ng * g.ldb;
  }
  const int kpad = (g.k + 3) & ~3;
  const uint32_t hrow = (uint32_t)(m0 + li);        // the dropout hash is keyed by the row of the BATCH, not of the source matrix

  f32x16 acc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  for (int c0 = 0; c0 < g.gpw; c0 += kGroups) {
    float4 av[kGroups], bv[NB][kGroups], sc[kGroups], sh[kGroups];
    // every load of the chunk first (clamped, always valid addresses: groups past K re-read the last float4 and are skipped below)
#pragma unroll
    for (int u = 0; u < kGroups; ++u) {
      const int kc = (q * g.gpw + c0 + u) * 8 + kk * 4;
      const int kcc = kc > kpad - 4 ? kpad - 4 : kc;
      av[u] = ld4(ap + kcc);
      if (XF == 0 && g.a_copy && blockIdx.y == 0 && c0 + u < g.gpw && kc < kpad && m0 + li < g.m)
        *reinterpret_cast<float4*>(g.a_copy + (m0 + li) * g.ld_copy + kc) = av[u];
      if (XF && !FIN) {
        sc[u] = ld4(g.a_scale + kcc);
        sh[u] = ld4(g.a_shift + kcc);
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        if (B_KN) {
          const int k0 = kc > g.k - 1 ? g.k - 1 : kc, k1 = kc + 1 > g.k - 1 ? g.k - 1 : kc + 1;
          const int k2 = kc + 2 > g.k - 1 ? g.k - 1 : kc + 2, k3 = kc + 3 > g.k - 1 ? g.k - 1 : kc + 3;
          bv[j][u].x = bp[j][(int64_t)k0 * g.ldb];
          bv[j][u].y = bp[j][(int64_t)k1 * g.ldb];
          bv[j][u].z = bp[j][(int64_t)k2 * g.ldb];
          bv[j][u].w = bp[j][(int64_t)k3 * g.ldb];
        } else if (!BU) {
          bv[j][u] = ld4(bp[j] + kcc);
        } else {                 // rows of W that start off a 16-byte boundary (cora: 1433 features): the lane's four k's one by one
          const int k0 = kc > g.k - 1 ? g.k - 1 : kc, k1 = kc + 1 > g.k - 1 ? g.k - 1 : kc + 1;
          const int k2 = kc + 2 > g.k - 1 ? g.k - 1 : kc + 2, k3 = kc + 3 > g.k - 1 ? g.k - 1 : kc + 3;
          bv[j][u].x = bp[j][k0];
          bv[j][u].y = bp[j][k1];
          bv[j][u].z = bp[j][k2];
          bv[j][u].w = bp[j][k3];
        }
      }
    }
    if (FIN) {
      if (c0 == 0) {              // the chunk's operand loads are in flight: combine the pending statistics of A's columns meanwhile
        for (int col = tid; col < g.k; col += 256) {
          const float gam = pend.gamma ? pend.gamma[col] : 1.f, bet = pend.beta ? pend.beta[col] : 0.f;
          double n = 0.0, sum = 0.0, qs = 0.0;
          for (int k0 = 0; k0 < pend.nparts; k0 += 16) {        // 32 loads in flight (clamped addresses), then the arithmetic
            float mk[16], m2k[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int kx = k0 + i < pend.nparts ? k0 + i : pend.nparts - 1;
              mk[i] = pend.ws_mean[(int64_t)kx * pend.pstride + col];
              m2k[i] = pend.ws_m2[(int64_t)kx * pend.pstride + col];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int64_t left = pend.rows - (int64_t)(k0 + i) * 32;       // rows of tile k0 + i (the partials are this file's 32-row tiles)
              const double nb = k0 + i < pend.nparts ? (double)(left < 32 ? left : 32) : 0.0;
              n += nb;
              sum += nb * (double)mk[i];
              qs += (k0 + i < pend.nparts ? (double)m2k[i] : 0.0) + nb * (double)mk[i] * (double)mk[i];
            }
          }
          const double mean = sum / n;
          double m2 = qs - sum * mean;
          if (m2 < 0.0) m2 = 0.0;
          if (col == 0 && blockIdx.x == 0 && blockIdx.y == 0 && pend.nbt) pend.nbt[0] += 1;
          bn_emit_column(pend, col, n, mean, m2, blockIdx.x == 0 && blockIdx.y == 0, gam, bet, s_sc[col], s_sh[col]);
        }
        __syncthreads();
      }
#pragma unroll
      for (int u = 0; u < kGroups; ++u) {
        const int kc = (q * g.gpw + c0 + u) * 8 + kk * 4;
        const int kcc = kc > kpad - 4 ? kpad - 4 : kc;
        sc[u] = ld4(&s_sc[kcc]);
        sh[u] = ld4(&s_sh[kcc]);
      }
    }
#pragma unroll
    for (int u = 0; u < kGroups; ++u) {
      const int kg = q * g.gpw + c0 + u;                  // wave-uniform
      if (c0 + u < g.gpw && kg * 8 < g.k) {
        const int kc = kg * 8 + kk * 4;
        const int kleft = g.k - kc;
        float4 a4 = av[u];
        if (XF) {
          a4.x = fmaxf(fmaf(a4.x, sc[u].x, sh[u].x), 0.f);
          a4.y = fmaxf(fmaf(a4.y, sc[u].y, sh[u].y), 0.f);
          a4.z = fmaxf(fmaf(a4.z, sc[u].z, sh[u].z), 0.f);
          a4.w = fmaxf(fmaf(a4.w, sc[u].w, sh[u].w), 0.f);
          if (XF == 2) {
            a4.x = glnn::drop_keep(g.drop_seed, g.drop_thr, hrow, (uint32_t)kc + 0) ? a4.x * g.drop_scale : 0.f;
            a4.y = glnn::drop_keep(g.drop_seed, g.drop_thr, hrow, (uint32_t)kc + 1) ? a4.y * g.drop_scale : 0.f;
            a4.z = glnn::drop_keep(g.drop_seed, g.drop_thr, hrow, (uint32_t)kc + 2) ? a4.z * g.drop_scale : 0.f;
            a4.w = glnn::drop_keep(g.drop_seed, g.drop_thr, hrow, (uint32_t)kc + 3) ? a4.w * g.drop_scale : 0.f;
          }
        }
        a4 = keep_first(a4, kleft);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const float4 b4 = keep_first(bv[j][u], kleft);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc[j], 0, 0, 0);
        }
      }
    }
  }

  // the four waves' partial tiles -> LDS; C fragment layout: register r of lane (li, kk) is row (r & 3) + 8 (r >> 2) + 4 kk, column li
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[q][((r & 3) + 8 * (r >> 2) + 4 * kk) * LDT + 32 * j + li] = acc[j][r];
  __syncthreads();
  // thread -> row tid >> 3, four columns at (tid & 7) * 4 of every 32-column block; quarters summed k ascending
  {
    const int row = tid >> 3;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int c4 = (tid & 7) * 4 + 32 * j;
      const float4 v0 = ld4(&red[0][row * LDT + c4]), v1 = ld4(&red[1][row * LDT + c4]);
      const float4 v2 = ld4(&red[2][row * LDT + c4]), v3 = ld4(&red[3][row * LDT + c4]);
      float v[4] = {((v0.x + v1.x) + v2.x) + v3.x, ((v0.y + v1.y) + v2.y) + v3.y, ((v0.z + v1.z) + v2.z) + v3.z,
                    ((v0.w + v1.w) + v2.w) + v3.w};
      const int col = n0 + c4;
      if (g.ep_scale) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int cc = col + t < g.n ? col + t : g.n - 1;
          v[t] = fmaf(v[t], g.ep_scale[cc], g.bias ? g.bias[cc] : 0.f);
        }
      } else if (g.bias) {
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] += g.bias[col + t < g.n ? col + t : g.n - 1];
      }
      if (g.relu) {
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.f);
      }
      float qv[4] = {0.f, 0.f, 0.f, 0.f};
      if (EPI == EPI_BNBWD) {                           // C -> dy (bn_dy of student.hip); qv = dy * xhat
        const float zz[4] = {ez.x, ez.y, ez.z, ez.w}, mu[4] = {emu.x, emu.y, emu.z, emu.w}, rs[4] = {ers.x, ers.y, ers.z, ers.w};
        const float sc4[4] = {esc.x, esc.y, esc.z, esc.w}, sh4[4] = {esh.x, esh.y, esh.z, esh.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float dav = v[t];
          if (g.e_thr) dav = glnn::drop_keep(g.e_seed, g.e_thr, (uint32_t)(m0 + row), (uint32_t)(col + t)) ? dav * g.e_dscale : 0.f;
          float dyv = (!g.e_relu || fmaf(zz[t], sc4[t], sh4[t]) > 0.f) ? dav : 0.f;
          if (m0 + row >= g.m || col + t >= g.n) dyv = 0.f;
          v[t] = dyv;
          qv[t] = dyv * ((zz[t] - mu[t]) * rs[t]);
        }
      }
      if (m0 + row < g.m) {
        float* cp = g.c + (m0 + row) * g.ldc + col;
        if (g.c_vec && col + 3 < g.n) {
          *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (col + t < g.n) cp[t] = v[t];
        }
      }
      if (EPI != EPI_PLAIN) *reinterpret_cast<float4*>(&red[0][row * LDT + c4]) = make_float4(v[0], v[1], v[2], v[3]);
      if (EPI == EPI_BNBWD) *reinterpret_cast<float4*>(&red[1][row * LDT + c4]) = make_float4(qv[0], qv[1], qv[2], qv[3]);
    }
  }
  if (EPI == EPI_PLAIN) return;
  __syncthreads();

  if (EPI == EPI_BNBWD) {
    // column sums of the tile's dy and dy * xhat (rows past m were zeroed): thread = (column c, 4-row group gq), fixed order
    __shared__ float p1[8][32], p2[8][32];
    const int c = tid & 31, gq = tid >> 5;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s1 += red[0][(gq * 4 + i) * LDT + c];
      s2 += red[1][(gq * 4 + i) * LDT + c];
    }
    p1[gq][c] = s1;
    p2[gq][c] = s2;
    __syncthreads();
    if (gq == 0 && n0 + c < g.n) {
      g.e_ws1[(int64_t)blockIdx.x * g.n + n0 + c] = ((p1[0][c] + p1[1][c]) + (p1[2][c] + p1[3][c])) + ((p1[4][c] + p1[5][c]) + (p1[6][c] + p1[7][c]));
      g.e_ws2[(int64_t)blockIdx.x * g.n + n0 + c] = ((p2[0][c] + p2[1][c]) + (p2[2][c] + p2[3][c])) + ((p2[4][c] + p2[5][c]) + (p2[6][c] + p2[7][c]));
    }
    return;
  }

  if (EPI == EPI_STATS) {
    // (mean, M2) of the tile's rows per column: thread = (column c, 4-row group gq); two passes over LDS
    __shared__ float ps[8][32];
    const int c = tid & 31, gq = tid >> 5;
    const int cnt = g.m - m0 < 32 ? (int)(g.m - m0) : 32;
    float x[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      x[i] = red[0][(gq * 4 + i) * LDT + c];
      s += gq * 4 + i < cnt ? x[i] : 0.f;
    }
    ps[gq][c] = s;
    __syncthreads();
    const float mean = (((ps[0][c] + ps[1][c]) + (ps[2][c] + ps[3][c])) + ((ps[4][c] + ps[5][c]) + (ps[6][c] + ps[7][c]))) / (float)cnt;
    __syncthreads();
    float qv = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float dlt = x[i] - mean;
      qv = gq * 4 + i < cnt ? fmaf(dlt, dlt, qv) : qv;
    }
    ps[gq][c] = qv;
    __syncthreads();
    if (gq == 0 && n0 + c < g.n) {
      const float m2 = ((ps[0][c] + ps[1][c]) + (ps[2][c] + ps[3][c])) + ((ps[4][c] + ps[5][c]) + (ps[6][c] + ps[7][c]));
      st_part(&g.ws_mean[(int64_t)blockIdx.x * g.n + n0 + c], mean, g.counters != nullptr);
      st_part(&g.ws_m2[(int64_t)blockIdx.x * g.n + n0 + c], m2, g.counters != nullptr);
    }
    if (!g.counters) return;                 // deferred: the next layer's launch combines the partials in its prologue (FIN)
    const int cb = n0 >> 6;                                         // 64-column block of the finalize; its 1 or 2 column tiles
    const int col_tiles = g.n - 64 * cb > 32 ? 2 : 1;
    if (last_workgroup(&g.counters[cb], (int)gridDim.x * col_tiles)) bn_finalize_columns<true>(fin, cb);
    return;
  }

  if (EPI == EPI_LOSS) {
    // rows of the tile are complete (n <= 64 = one tile).  All 32 rows at once: 8 threads per row, 8 classes per thread, the row
    // reductions are 3 xor-shuffle steps inside the 8-lane group (a wave per row walked 8 rows x 4 reductions x 6 steps in sequence:
    // 22 us for this kernel).  Same formulas as softmax_loss_kernel; the sums over classes associate differently.
    const int r = tid >> 3, j8 = (tid & 7) * 8;
    const int64_t row = m0 + r;
    float z[8], gk[8];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      z[t] = j8 + t < ls.c ? red[0][r * LDT + j8 + t] : -INFINITY;
      mx = fmaxf(mx, z[t]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 1)); mx = fmaxf(mx, __shfl_xor(mx, 2)); mx = fmaxf(mx, __shfl_xor(mx, 4));
    float se = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) se += j8 + t < ls.c ? expf(z[t] - mx) : 0.f;
    se += __shfl_xor(se, 1); se += __shfl_xor(se, 2); se += __shfl_xor(se, 4);
    const float lse = mx + logf(se);
    float row_loss = 0.f;
    if (ls.kind == GLNN_LOSS_NLL) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const float lp = z[t] - lse;
        const bool hit = (int64_t)(j8 + t) == yv;
        gk[t] = (expf(lp) - (hit ? 1.f : 0.f)) * ls.scale;
        row_loss += (hit && j8 + t < ls.c) ? -lp : 0.f;
      }
    } else {
      float et[8];
      float set = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        et[t] = j8 + t < ls.c ? expf(tv[t]) : 0.f;
        set += et[t];
      }
      set += __shfl_xor(set, 1); set += __shfl_xor(set, 2); set += __shfl_xor(set, 4);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const float lp = z[t] - lse;
        row_loss += j8 + t < ls.c ? et[t] * (tv[t] - lp) : 0.f;
        gk[t] = (expf(lp) * set - et[t]) * ls.scale;
      }
    }
    row_loss += __shfl_xor(row_loss, 1); row_loss += __shfl_xor(row_loss, 2); row_loss += __shfl_xor(row_loss, 4);
    const bool row_ok = row < g.m;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const bool ok = row_ok && j8 + t < ls.c;
      if (ok) ls.dz[row * ls.ldg + j8 + t] = gk[t];
      red[1][r * LDT + j8 + t] = ok ? gk[t] : 0.f;         // for the column sums (the bias gradient) below
    }
    __shared__ float rl[32];
    __shared__ float sc4[4][64];
    if ((tid & 7) == 0) rl[r] = row_ok ? row_loss : 0.f;
    __syncthreads();
    if (tid == 0) {
      float l = 0.f;
#pragma unroll
      for (int i = 0; i < 32; ++i) l += rl[i];
      st_part(&ls.partial[blockIdx.x], l, !ls.defer);
    }
    if (ls.col_sum && tid < 64) {
      float cs = 0.f;
#pragma unroll
      for (int i = 0; i < 32; ++i) cs += red[1][i * LDT + tid];
      st_part(&ls.col_partial[(int64_t)blockIdx.x * 64 + tid], cs, !ls.defer);
    }
    if (ls.defer) return;                   // the fused Adam launch folds the partials
    if (!last_workgroup(ls.counter, (int)gridDim.x)) return;
    loss_fold_last(ls, (int)gridDim.x, sc4);
  }
}

template <int NB, int EPI>
int launch_lat(const LatArgs& g, const BnFinArgs& fin, const LossArgs& ls, const BnFinArgs* pend, bool b_kn, hipStream_t st) {
  const dim3 grid((unsigned)((g.m + 31) / 32), (unsigned)((g.n + 32 * NB - 1) / (32 * NB)));
  const int xf = (!g.a_scale && !pend) ? 0 : (g.drop_thr ? 2 : 1);
  const BnFinArgs none = {};
#define GLNN_LAT_LAUNCH(XF_, KN_, FIN_) \
  hipLaunchKernelGGL((gemm_lat_kernel<XF_, KN_, NB, EPI, FIN_>), grid, dim3(256), 0, st, g, fin, ls, pend ? *pend : none)
  if (b_kn) {
    if (xf != 0 || (EPI != EPI_PLAIN && EPI != EPI_BNBWD)) return GLNN_ERR_UNSUPPORTED;       // the input-gradient product: plain operands
    if constexpr (EPI == EPI_PLAIN || EPI == EPI_BNBWD) GLNN_LAT_LAUNCH(0, true, false);
  } else if (EPI == EPI_BNBWD) {
    return GLNN_ERR_UNSUPPORTED;
  } else if (!g.b_vec) {
    if (pend || xf != 0 || NB != 1 || EPI == EPI_LOSS) return GLNN_ERR_UNSUPPORTED;   // unaligned W: first layers (plain A, hidden output) only
    if constexpr (NB == 1 && EPI != EPI_LOSS)
      hipLaunchKernelGGL((gemm_lat_kernel<0, false, NB, EPI, false, true>), grid, dim3(256), 0, st, g, fin, ls, none);
  } else if (pend) {
    if (xf == 1) GLNN_LAT_LAUNCH(1, false, true); else GLNN_LAT_LAUNCH(2, false, true);
  } else {
    if (xf == 0) GLNN_LAT_LAUNCH(0, false, false); else if (xf == 1) GLNN_LAT_LAUNCH(1, false, false); else GLNN_LAT_LAUNCH(2, false, false);
  }
#undef GLNN_LAT_LAUNCH
  return glnn::check_launch("glnn::gemm_lat");
}

// ---------------------------------------------------------------------------------------------
// The weight gradients of a small step in the same style: C_p[i, j] = sum_m A_p[m, i] * B'_p[m, j] for up to four problems in one
// launch (the batched form of gemm.hip's gemm_tn_multi_kernel, which runs 64 x 64 tiles through LDS with a transposing loader and
// splits the reduction over workgroups -- slabs + a fold).  Both operands are row-major along the NON-reduction index, which is exactly
// what a lane of v_mfma_f32_32x32x2 wants when it loads from global memory itself: lane (li, kk) needs A[m][i0 + li] for its m's -- 32
// lanes read 128 consecutive bytes.  So: a workgroup owns a 32 x 32 tile of one C_p, its four waves split the reduction (the batch
// rows), each wave keeps 8 k-groups (64 dword loads) in flight, the four partial tiles are summed through LDS in fixed order and C is
// written once: no slabs, no fold launch, no transposes.  B' = B, or dropout(relu(B * scale[j] + shift[j])) with the per-COLUMN
// constants in two registers of the lane.
// ---------------------------------------------------------------------------------------------
constexpr int kTnLatMax = 4;
struct TnLatProblem {
  const float* a; int64_t lda; const float* b; int64_t ldb; const float* b_scale; const float* b_shift;
  uint32_t drop_thr, drop_seed; float drop_scale;
  int m, ka, nb; float* c; int64_t ldc;
  int gj;          // 32-column tiles per row of tiles
  int start;       // first workgroup of this problem
  int gpw;         // k-groups (8 batch rows) per wave
  int splits;      // > 1: the batch rows are also split over `splits` workgroups per tile; split s writes slab c + s * slab (c = workspace)
  int64_t slab;
  // bn_z != NULL: `a` is dy and the operand is the BatchNorm backward's apply, dz = gamma * rstd * (dy - S1/B - xhat * S2/B), evaluated in
  // the loads (bn_apply_tiles_kernel's expression and partial order: dz is never stored).  Column i of A is ONE lane's, so its constants
  // and the fold of the S1 / S2 tile partials are per-lane scalars.  The j0 == 0 tiles also leave the column sums of dz per (split, wave)
  // in bn_ws3 [4 * splits][ka] (the bias gradient in front of the BatchNorm; Adam folds them); split 0 / wave 0 stores dgamma, dbeta.
  const float* bn_z; int64_t bn_ldz; const float* bn_gamma; const float* bn_mean; const float* bn_rstd; const float* bn_p1; const float* bn_p2;
  int bn_nparts; float* bn_dgamma; float* bn_dbeta; float* bn_ws3;
};
struct TnLatArgs { TnLatProblem p[kTnLatMax]; int n; };

__global__ __launch_bounds__(256) void gemm_tn_lat_kernel(const TnLatArgs args) {
  int pi = 0;
#pragma unroll
  for (int q = 1; q < kTnLatMax; ++q)
    if (q < args.n && (int)blockIdx.x >= args.p[q].start) pi = q;
  const TnLatProblem& g = args.p[pi];
  constexpr int LDT = 36;
  __shared__ __attribute__((aligned(16))) float red[4][32 * LDT];
  const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6, li = lane & 31, kk = lane >> 5;
  const int tile = ((int)blockIdx.x - g.start) / g.splits, split = ((int)blockIdx.x - g.start) % g.splits;
  const int i0 = (tile / g.gj) * 32, j0 = (tile % g.gj) * 32;
  const int wq = split * 4 + q;                      // this wave's slice of the reduction
  const int ic = i0 + li < g.ka ? i0 + li : g.ka - 1, jc = j0 + li < g.nb ? j0 + li : g.nb - 1;
  const float* ap = g.a + ic;
  const float* bp = g.b + jc;
  const bool xf = g.b_scale != nullptr;
  const float sc = xf ? g.b_scale[jc] : 1.f, sh = xf ? g.b_shift[jc] : 0.f;
  const bool bn = g.bn_z != nullptr;
  float mu = 0.f, rs = 1.f, grs = 1.f, c1 = 0.f, c2 = 0.f, cs = 0.f;
  const float* zp = bn ? g.bn_z + ic : nullptr;
  if (bn) {
    float S1 = 0.f, S2 = 0.f;
    for (int k0 = 0; k0 < g.bn_nparts; k0 += 16) {          // bn_apply_tiles_kernel's fold: k ascending, 16 partials of each in flight
      float q1[16], q2[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int k = k0 + u < g.bn_nparts ? k0 + u : g.bn_nparts - 1;
        q1[u] = g.bn_p1[(int64_t)k * g.ka + ic];
        q2[u] = g.bn_p2[(int64_t)k * g.ka + ic];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (k0 + u < g.bn_nparts) { S1 += q1[u]; S2 += q2[u]; }
    }
    mu = g.bn_mean[ic]; rs = g.bn_rstd[ic];
    grs = g.bn_gamma[ic] * rs;
    const float inv_b = 1.0f / (float)g.m;
    c1 = S1 * inv_b; c2 = S2 * inv_b;
    if (j0 == 0 && split == 0 && q == 0 && kk == 0 && i0 + li < g.ka) {
      g.bn_dbeta[ic] = S1;
      g.bn_dgamma[ic] = S2;
    }
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int c0 = 0; c0 < g.gpw; c0 += kGroups) {
    float av[kGroups][4], bv[kGroups][4], zv[kGroups][4];
#pragma unroll
    for (int u = 0; u < kGroups; ++u) {
      const int mb = (wq * g.gpw + c0 + u) * 8 + kk * 4;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int64_t mr = mb + t < g.m ? mb + t : g.m - 1;          // clamped: always a valid row, masked below
        av[u][t] = ap[mr * g.lda];
        bv[u][t] = bp[mr * g.ldb];
        zv[u][t] = bn ? zp[mr * g.bn_ldz] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < kGroups; ++u) {
      const int kg = wq * g.gpw + c0 + u;                             // wave-uniform
      if (c0 + u < g.gpw && kg * 8 < g.m) {
        const int mb = kg * 8 + kk * 4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float bvv = bv[u][t];
          if (xf) {
            bvv = fmaxf(fmaf(bvv, sc, sh), 0.f);
            if (g.drop_thr) bvv = glnn::drop_keep(g.drop_seed, g.drop_thr, (uint32_t)(mb + t), (uint32_t)jc) ? bvv * g.drop_scale : 0.f;
          }
          const bool in = mb + t < g.m;
          float avv = av[u][t];
          if (bn) {
            avv = in ? glnn::bn_dz(grs, avv, c1, zv[u][t], mu, rs, c2) : 0.f;
            cs += avv;
          }
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(in ? avv : 0.f, in ? bvv : 0.f, acc, 0, 0, 0);
        }
      }
    }
  }
  if (bn) {                       // this wave's column sums of dz: the two lane halves hold different batch rows of the same column
    cs += __shfl_xor(cs, 32);
    if (j0 == 0 && kk == 0 && i0 + li < g.ka) g.bn_ws3[(int64_t)(split * 4 + q) * g.ka + ic] = cs;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[q][((r & 3) + 8 * (r >> 2) + 4 * kk) * LDT + li] = acc[r];
  __syncthreads();
  const int row = tid >> 3, c4 = (tid & 7) * 4;
  const float4 v0 = ld4(&red[0][row * LDT + c4]), v1 = ld4(&red[1][row * LDT + c4]);
  const float4 v2 = ld4(&red[2][row * LDT + c4]), v3 = ld4(&red[3][row * LDT + c4]);
  const float v[4] = {((v0.x + v1.x) + v2.x) + v3.x, ((v0.y + v1.y) + v2.y) + v3.y, ((v0.z + v1.z) + v2.z) + v3.z, ((v0.w + v1.w) + v2.w) + v3.w};
  if (i0 + row < g.ka) {
    float* cp = g.c + (int64_t)split * g.slab + (int64_t)(i0 + row) * g.ldc + j0 + c4;
    if (j0 + c4 + 3 < g.nb && g.ldc % 4 == 0 && glnn::aligned16(g.c)) {
      *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (j0 + c4 + t < g.nb) cp[t] = v[t];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The BatchNorm / ReLU / dropout backward of a small step without a wait inside a launch: the column sums S1 = sum dy, S2 = sum dy * xhat
// come as per-tile partials from the epilogue of the input-gradient GEMM above (EPI_BNBWD, which also stored dy); this kernel folds the
// partials of its 64 columns while its dy / z loads are in flight (k ascending: every workgroup the same bits), applies
//   dz = gamma * rstd * (dy - S1/B - xhat * S2/B)
// and leaves the per-chunk column sums of dz (the bias gradient of the Linear in front) as partials for the Adam launch or a fold
// launch; chunk 0 stores dgamma = S2, dbeta = S1.  Grid (64-column blocks, 32-row chunks); thread = (column, row lane of 8 rows).
// ---------------------------------------------------------------------------------------------
struct BnApplyArgs {
  const float* dy; int64_t lddy; const float* z; int64_t ldz; int64_t rows; int h;
  const float* gamma; const float* mean; const float* rstd; const float* p1; const float* p2; int nparts;
  float* dz; int64_t lddz; float* dgamma; float* dbeta; float* ws3;
};
__global__ __launch_bounds__(256) void bn_apply_tiles_kernel(const BnApplyArgs a) {
  const int lc = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lc;
  const int colc = col < a.h ? col : a.h - 1;
  const int64_t r0 = (int64_t)blockIdx.y * 32;
  float dyv[8], zv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t r = r0 + rl + 4 * i;
    const int64_t rc = r < a.rows ? r : a.rows - 1;
    dyv[i] = a.dy[rc * a.lddy + colc];
    zv[i] = a.z[rc * a.ldz + colc];
  }
  const float mu = a.mean[colc], rs = a.rstd[colc], g = a.gamma[colc];
  float S1 = 0.f, S2 = 0.f;
  for (int k0 = 0; k0 < a.nparts; k0 += 16) {
    float q1[16], q2[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int k = k0 + u < a.nparts ? k0 + u : a.nparts - 1;
      q1[u] = a.p1[(int64_t)k * a.h + colc];
      q2[u] = a.p2[(int64_t)k * a.h + colc];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (k0 + u < a.nparts) { S1 += q1[u]; S2 += q2[u]; }
  }
  if (blockIdx.y == 0 && rl == 0 && col < a.h) {
    a.dbeta[col] = S1;
    a.dgamma[col] = S2;
  }
  const float inv_b = 1.0f / (float)a.rows;
  const float c1 = S1 * inv_b, c2 = S2 * inv_b, grs = g * rs;
  float sdz = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t r = r0 + rl + 4 * i;
    if (r < a.rows) {
      const float out = glnn::bn_dz(grs, dyv[i], c1, zv[i], mu, rs, c2);
      if (col < a.h) a.dz[r * a.lddz + col] = out;
      sdz += out;
    }
  }
  __shared__ float sh[4][64];
  sh[rl][lc] = sdz;
  __syncthreads();
  if (rl == 0 && col < a.h && a.ws3) a.ws3[(int64_t)blockIdx.y * a.h + col] = (sh[0][lc] + sh[1][lc]) + (sh[2][lc] + sh[3][lc]);
}
__global__ void tile_chunk_sum_kernel(const float* __restrict__ ws, int nchunks, int h, float* __restrict__ out) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= h) return;
  float s = 0.f;
#pragma unroll 8
  for (int k = 0; k < nchunks; ++k) s += ws[(int64_t)k * h + col];
  out[col] = s;
}

__global__ __launch_bounds__(256) void bn_finalize_tiles_kernel(const BnFinArgs a) { bn_finalize_columns<false>(a, blockIdx.x); }


BnFinArgs fin_args(const glnn::LatStats& st, const float* ws_mean, const float* ws_m2, int64_t mt, int64_t m, int n) {
  BnFinArgs fin = {};
  fin.ws_mean = ws_mean; fin.ws_m2 = ws_m2; fin.nparts = (int)mt; fin.pstride = n; fin.rows = m; fin.h = n; fin.chunk_rows = 32;
  fin.gamma = st.gamma; fin.beta = st.beta; fin.eps = st.eps; fin.momentum = st.momentum; fin.running_mean = st.running_mean;
  fin.running_var = st.running_var; fin.nbt = st.nbt; fin.mean_out = st.mean_out; fin.rstd_out = st.rstd_out;
  fin.a_scale = st.a_scale_out; fin.a_shift = st.a_shift_out;
  return fin;
}

}  // namespace

// C[m,n] = A'[m,k] * B + bias with an optional fused epilogue (st: BatchNorm statistics | ls: loss + dlogits); at most one of st / ls.
//   st->counters != NULL: statistics finished inside the launch (last workgroup); NULL: DEFERRED -- only the per-tile partials are
//   written (st->ws: 2 * ceil(m/32) * n floats) and the launch that consumes C must be given the same LatStats as `pend`
//   (or glnn::bn_finalize_tiles run on it).
//   pend: A's columns carry deferred statistics (a_scale / a_shift arguments are ignored; pend->ws holds the partials of A = the C of
//   the producing call with the same m).
// GLNN_ERR_UNSUPPORTED (nothing launched, no error text) when the problem is outside the latency regime or an operand is not
// float4-addressable: the caller then issues the tiled GEMM and the separate reduction kernels.
int glnn::gemm_lat(const float* a, int64_t lda, const int64_t* a_rows, const float* a_scale, const float* a_shift, float drop_p,
                   uint32_t drop_seed, int64_t m, int k, const float* b, int64_t ldb, int b_layout, int n, const float* bias, float* c,
                   int64_t ldc, const LatStats* pend, const LatStats* st, const LatLoss* ls, void* stream, float* a_copy, int64_t ld_copy,
                   const float* ep_scale, int relu) {
  const int enabled = glnn::opts().gemm_lat;
  constexpr int max_m = 1024, max_k = 256, max_n = 512;      // the latency regime (sweeps of rounds 2-3)
  // W[n, k] whose rows are not float4-addressable (a feature width that is not a multiple of 4: cora's 1433) would take the guarded
  // generic GEMM (206 us for 140 x 128 x 1433): here it is loaded dword by dword, and K may be as deep as it comes
  const bool b_vec = b_layout || (ldb % 4 == 0 && glnn::aligned16(b));
  const int k_lim = b_vec ? max_k : (max_k > (1 << 16) ? max_k : (1 << 16));   // (penn94: 4814 features; the alternative runs at 1.6 TF)
  const int64_t m_lim = b_vec ? max_m : ((int64_t)1 << 28);          // (the alternative for unaligned W is a kernel of 2 workgroups per 128 x 128 tile)
  if (!enabled || !a || !b || !c || m < 1 || m > m_lim || k < 4 || k > k_lim || n < 1 || n > max_n || (st && ls)) return GLNN_ERR_UNSUPPORTED;
  const int kpad = (k + 3) & ~3;
  if (lda % 4 || !glnn::aligned16(a) || lda < kpad) return GLNN_ERR_UNSUPPORTED;
  if (!b_layout && ldb < (b_vec ? kpad : k)) return GLNN_ERR_UNSUPPORTED;
  if (b_layout && ldb < n) return GLNN_ERR_UNSUPPORTED;
  if (!b_vec && (pend || a_scale || ls)) return GLNN_ERR_UNSUPPORTED;          // the dword-loading variant exists for plain first layers only
  const int64_t mt = (m + 31) / 32;
  if (pend) {
    if (b_layout || k % 4 || k > kFinMaxK || !pend->ws || pend->ws_floats < 2 * mt * k || !pend->a_scale_out || !pend->a_shift_out) return GLNN_ERR_UNSUPPORTED;
  } else {
    if ((a_scale == nullptr) != (a_shift == nullptr)) return GLNN_ERR_UNSUPPORTED;
    if (a_scale && (k % 4 || !glnn::aligned16(a_scale) || !glnn::aligned16(a_shift))) return GLNN_ERR_UNSUPPORTED;
    if (drop_p > 0.f && !a_scale) return GLNN_ERR_UNSUPPORTED;
  }
  LatArgs g = {};
  g.a = a; g.lda = lda; g.a_rows = a_rows; g.a_scale = pend ? nullptr : a_scale; g.a_shift = pend ? nullptr : a_shift;
  g.drop_thr = glnn::drop_threshold(drop_p); g.drop_seed = drop_seed; g.drop_scale = 1.0f / (1.0f - drop_p);
  g.m = m; g.k = k; g.b = b; g.ldb = ldb; g.n = n; g.bias = bias; g.c = c; g.ldc = ldc;
  g.c_vec = (ldc % 4 == 0) && glnn::aligned16(c);
  g.b_vec = b_vec ? 1 : 0;
  if ((ep_scale || relu) && (st || ls)) return GLNN_ERR_UNSUPPORTED;
  g.ep_scale = ep_scale; g.relu = relu ? 1 : 0;
  if (a_copy) {
    if (a_scale || pend || ld_copy % 4 || ld_copy < kpad || !glnn::aligned16(a_copy)) return GLNN_ERR_UNSUPPORTED;
    g.a_copy = a_copy; g.ld_copy = ld_copy;
  }
  const int groups = (k + 7) / 8;
  g.gpw = (groups + 3) / 4;
  BnFinArgs fin = {}, pfin = {};
  if (pend) pfin = fin_args(*pend, pend->ws, pend->ws + mt * k, mt, m, k);
  LossArgs la = {};
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (st) {
    if (!st->ws || st->ws_floats < 2 * mt * n || !st->a_scale_out || !st->a_shift_out || (n + 63) / 64 > 512) return GLNN_ERR_UNSUPPORTED;
    g.ws_mean = st->ws; g.ws_m2 = st->ws + mt * n; g.counters = st->counters;
    fin = fin_args(*st, g.ws_mean, g.ws_m2, mt, m, n);
    return launch_lat<1, EPI_STATS>(g, fin, la, pend ? &pfin : nullptr, b_layout != 0, s);
  }
  if (ls) {
    if (n > 64 || !ls->counter || !ls->ws || ls->ws_floats < mt * 65 || !ls->dlogits || ls->ldg < n) return GLNN_ERR_UNSUPPORTED;
    if (ls->kind == GLNN_LOSS_NLL ? !ls->labels : (ls->kind != GLNN_LOSS_KL || !ls->target_logp || ls->ldt < n)) return GLNN_ERR_UNSUPPORTED;
    la.z = c; la.ldz = ldc; la.rows = m; la.c = n; la.kind = ls->kind; la.labels = ls->labels; la.label_rows = ls->label_rows;
    la.t = ls->target_logp; la.ldt = ls->ldt; la.t_rows = ls->target_rows; la.scale = ls->lamb / (float)m;
    la.dz = ls->dlogits; la.ldg = ls->ldg; la.partial = ls->ws; la.counter = ls->counter; la.inv_rows = 1.0f / (float)m;
    la.loss_out = ls->loss_out; la.loss_accum = ls->loss_accum; la.col_sum = ls->col_sum; la.col_partial = ls->ws + mt;
    const bool defer = ls->pf && ls->pf->n < glnn::kMaxGradFolds;
    la.defer = defer ? 1 : 0;
    const int rc = launch_lat<2, EPI_LOSS>(g, fin, la, pend ? &pfin : nullptr, b_layout != 0, s);
    if (rc == GLNN_OK && defer) {          // registered only once the launch is in the queue: an UNSUPPORTED variant must leave *pf alone
      ls->pf->has_loss = 1;
      ls->pf->loss = {ls->ws, (int)mt, 1.0f / (float)m, ls->loss_out, ls->loss_accum};
      if (ls->col_sum) ls->pf->e[ls->pf->n++] = {ls->col_sum, la.col_partial, (int)mt, 1, 64};
    }
    return rc;
  }
  return launch_lat<1, EPI_PLAIN>(g, fin, la, pend ? &pfin : nullptr, b_layout != 0, s);
}

// the deferred statistics of gemm_lat finished by a launch of their own (the consumer turned out not to be a latency GEMM)
int glnn::bn_finalize_tiles(const LatStats& st, int64_t m, int n, void* stream) {
  const int64_t mt = (m + 31) / 32;
  GLNN_REQUIRE(st.ws && st.ws_floats >= 2 * mt * n && st.a_scale_out && st.a_shift_out, "glnn::bn_finalize_tiles: bad arguments");
  const BnFinArgs fin = fin_args(st, st.ws, st.ws + mt * n, mt, m, n);
  hipLaunchKernelGGL(bn_finalize_tiles_kernel, dim3((n + 63) / 64), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), fin);
  return glnn::check_launch("glnn::bn_finalize_tiles");
}

// Up to four weight gradients (arguments as gemm_tn_batch) in ONE launch of the latency kernel, each written straight to its C (no
// slabs, no fold).  GLNN_ERR_UNSUPPORTED (nothing launched) unless every problem is small (m <= 1024, ka, nb <= 256), ungathered and
// -- with an operand transform -- has its column constants readable.
// defer + workspace (the fused Adam launch follows): long reductions are ALSO split over workgroups so that a wave runs one 8-group
// chunk; the slabs go to the workspace and defer[p] tells Adam how to fold them (k ascending).
int glnn::gemm_tn_lat(const TnProblem* pr, int n, void* stream, GradFold* defer, float* workspace, int64_t workspace_floats) {
  if (!glnn::opts().gemm_tn_lat || n < 1 || n > kTnLatMax) return GLNN_ERR_UNSUPPORTED;
  constexpr int max_dim = 256;
  TnLatArgs a = {};
  a.n = n;
  int blocks = 0;
  int64_t ws_off = 0;
  GradFold extra[2];            // column-sum folds of problems with a BatchNorm apply in their operand: reported behind defer[n - 1]
  int n_extra = 0;
  for (int p = 0; p < n; ++p) {
    const TnProblem& q = pr[p];
    if (!(q.a && q.b && q.c) || q.b_rows || q.m < 1 || q.m > 1024 || q.ka < 1 || q.nb < 1 || q.ka > max_dim || q.nb > max_dim) return GLNN_ERR_UNSUPPORTED;
    if (q.lda < q.ka || q.ldb < q.nb || q.ldc < q.nb) return GLNN_ERR_UNSUPPORTED;
    if ((q.b_scale == nullptr) != (q.b_shift == nullptr) || q.drop_p < 0.f || q.drop_p >= 1.f || (q.drop_p > 0.f && !q.b_scale)) return GLNN_ERR_UNSUPPORTED;
    TnLatProblem& g = a.p[p];
    g.a = q.a; g.lda = q.lda; g.b = q.b; g.ldb = q.ldb; g.b_scale = q.b_scale; g.b_shift = q.b_shift;
    g.drop_thr = glnn::drop_threshold(q.drop_p); g.drop_seed = q.drop_seed; g.drop_scale = 1.0f / (1.0f - q.drop_p);
    g.m = (int)q.m; g.ka = q.ka; g.nb = q.nb; g.c = q.c; g.ldc = q.ldc;
    const int gi = (q.ka + 31) / 32;
    g.gj = (q.nb + 31) / 32;
    g.start = blocks;
    const int groups = (int)((q.m + 7) / 8);
    g.splits = 1; g.slab = 0;
    if (defer) defer[p] = {q.c, nullptr, 0, 0, 0};
    const int64_t slab = (int64_t)q.ka * q.nb;
    const int gpw_target = kGroups;
    int sp = (groups + 4 * gpw_target - 1) / (4 * gpw_target);
    const int max_sp = glnn::opts().gemm_tn_lat_splits;               // 1 = the unsplit form (bit-identical to the two-call step)
    if (sp > max_sp) sp = max_sp;
    if (defer && workspace && sp > 1 && q.ldc == q.nb) {
      ws_off = (ws_off + 3) & ~(int64_t)3;
      if (ws_off + sp * slab <= workspace_floats) {
        g.splits = sp; g.slab = slab; g.c = workspace + ws_off; g.ldc = q.nb;
        defer[p] = {q.c, g.c, sp, 0, slab};
        ws_off += sp * slab;
      }
    }
    g.gpw = (groups + 4 * g.splits - 1) / (4 * g.splits);
    blocks += gi * g.gj * g.splits;
    if (q.bn_z) {            // the operand is an un-applied BatchNorm backward: only with Adam next (its column sums are folded there)
      if (!defer || !workspace || !q.bn_gamma || !q.bn_mean || !q.bn_rstd || !q.bn_p1 || !q.bn_p2 || q.bn_nparts < 1 || !q.bn_dgamma || !q.bn_dbeta ||
          !q.bn_colsum || q.bn_ldz < q.ka || n_extra >= 2)
        return GLNN_ERR_UNSUPPORTED;
      ws_off = (ws_off + 3) & ~(int64_t)3;
      if (ws_off + 4 * (int64_t)g.splits * q.ka > workspace_floats) return GLNN_ERR_UNSUPPORTED;
      g.bn_z = q.bn_z; g.bn_ldz = q.bn_ldz; g.bn_gamma = q.bn_gamma; g.bn_mean = q.bn_mean; g.bn_rstd = q.bn_rstd; g.bn_p1 = q.bn_p1; g.bn_p2 = q.bn_p2;
      g.bn_nparts = q.bn_nparts; g.bn_dgamma = q.bn_dgamma; g.bn_dbeta = q.bn_dbeta; g.bn_ws3 = workspace + ws_off;
      extra[n_extra++] = {q.bn_colsum, g.bn_ws3, 4 * g.splits, 0, (int64_t)q.ka};
      ws_off += 4 * (int64_t)g.splits * q.ka;
    }
  }
  if (defer) for (int e = 0; e < 2; ++e) defer[n + e] = e < n_extra ? extra[e] : GradFold{nullptr, nullptr, 0, 0, 0};
  for (int p = n; p < kTnLatMax; ++p) a.p[p].start = blocks;
  hipLaunchKernelGGL(gemm_tn_lat_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
  return glnn::check_launch("glnn::gemm_tn_lat");
}

// The input gradient of a hidden layer followed by its BatchNorm / ReLU / dropout backward as TWO launches without a wait between
// workgroups: da = dz_up[m, k] * W[k, n] on the latency kernel whose epilogue turns it into dy (stored to `da`) and leaves the tile
// partials of sum dy / sum dy * xhat in `workspace` (3 * ceil(m/32) * n floats: S1, S2, then the chunk sums of dz), then
// bn_apply_tiles_kernel.  The bias-gradient column sums are folded by a launch here, or -- defer_colsum -- left for the fused Adam
// launch (k ascending).  GLNN_ERR_UNSUPPORTED (nothing launched) outside the latency regime.
int glnn::lat_dgrad_bn_bwd(const float* dz_up, int64_t ld_up, int64_t m, int k, const float* w, int64_t ldw, int n, const float* z, int64_t ldz,
                           const float* gamma, const float* mean, const float* rstd, const float* a_scale, const float* a_shift, float drop_p,
                           uint32_t drop_seed, float* da, int64_t ldda, float* dz, int64_t lddz, float* dgamma, float* dbeta, float* dz_col_sum,
                           float* workspace, int64_t workspace_floats, void* stream, GradFold* defer_colsum, int skip_apply) {
  constexpr int max_m = 1024, max_n = 1024, max_k = 1024;
  if (!glnn::opts().gemm_lat || !glnn::opts().lat_bn_bwd) return GLNN_ERR_UNSUPPORTED;
  if (!(dz_up && w && z && gamma && mean && rstd && a_scale && a_shift && da && dz && dgamma && dbeta && workspace)) return GLNN_ERR_UNSUPPORTED;
  if (m < 1 || m > max_m || k < 4 || k > max_k || n < 4 || n > max_n || n % 4) return GLNN_ERR_UNSUPPORTED;
  const int kpad = (k + 3) & ~3;
  if (ld_up % 4 || !glnn::aligned16(dz_up) || ld_up < kpad || ldw < n || ldz % 4 || !glnn::aligned16(z) || ldz < n) return GLNN_ERR_UNSUPPORTED;
  if (!glnn::aligned16(mean) || !glnn::aligned16(rstd) || !glnn::aligned16(a_scale) || !glnn::aligned16(a_shift)) return GLNN_ERR_UNSUPPORTED;
  if (ldda < n || lddz < n || drop_p < 0.f || drop_p >= 1.f) return GLNN_ERR_UNSUPPORTED;
  const int64_t mt = (m + 31) / 32;
  if (workspace_floats < 3 * mt * n) return GLNN_ERR_UNSUPPORTED;
  LatArgs g = {};
  g.a = dz_up; g.lda = ld_up; g.m = m; g.k = k; g.b = w; g.ldb = ldw; g.n = n; g.c = da; g.ldc = ldda;
  g.c_vec = (ldda % 4 == 0) && glnn::aligned16(da);
  g.drop_scale = 1.f;
  g.gpw = ((k + 7) / 8 + 3) / 4;
  g.e_z = z; g.e_ldz = ldz; g.e_mean = mean; g.e_rstd = rstd; g.e_sc = a_scale; g.e_sh = a_shift;
  g.e_thr = glnn::drop_threshold(drop_p); g.e_seed = drop_seed; g.e_dscale = 1.0f / (1.0f - drop_p); g.e_relu = 1;
  g.e_ws1 = workspace; g.e_ws2 = workspace + mt * n;
  const BnFinArgs none = {};
  const LossArgs nol = {};
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int rc = launch_lat<1, EPI_BNBWD>(g, none, nol, nullptr, true, st);
  if (rc != GLNN_OK || skip_apply) return rc;          // skip_apply: the caller's weight-gradient launch applies (TnProblem::bn_z) or calls bn_apply_tiles
  return glnn::bn_apply_tiles(da, ldda, z, ldz, m, n, gamma, mean, rstd, dz, lddz, dgamma, dbeta, dz_col_sum, workspace, workspace_floats, stream,
                              defer_colsum);
}

// the apply half of lat_dgrad_bn_bwd on its own (workspace = the one the GEMM's epilogue filled: S1 partials, S2 partials, room for ws3)
int glnn::bn_apply_tiles(const float* dy, int64_t lddy, const float* z, int64_t ldz, int64_t m, int n, const float* gamma, const float* mean,
                         const float* rstd, float* dz, int64_t lddz, float* dgamma, float* dbeta, float* dz_col_sum, float* workspace,
                         int64_t workspace_floats, void* stream, GradFold* defer_colsum) {
  const int64_t mt = (m + 31) / 32;
  GLNN_REQUIRE(dy && z && gamma && mean && rstd && dz && dgamma && dbeta && workspace && workspace_floats >= 3 * mt * n, "glnn::bn_apply_tiles: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  BnApplyArgs a = {dy, lddy, z, ldz, m, n, gamma, mean, rstd, workspace, workspace + mt * n, (int)mt, dz, lddz, dgamma, dbeta,
                   dz_col_sum ? workspace + 2 * mt * n : nullptr};
  hipLaunchKernelGGL(bn_apply_tiles_kernel, dim3((n + 63) / 64, (unsigned)mt), dim3(256), 0, st, a);
  if (dz_col_sum) {
    if (defer_colsum && mt <= 32) *defer_colsum = {dz_col_sum, a.ws3, (int)mt, 0, (int64_t)n};
    else {
      if (defer_colsum) *defer_colsum = {dz_col_sum, nullptr, 0, 0, 0};
      hipLaunchKernelGGL(tile_chunk_sum_kernel, dim3((n + 127) / 128), dim3(128), 0, st, a.ws3, (int)mt, n, dz_col_sum);
    }
  }
  return glnn::check_launch("glnn::bn_apply_tiles");
}
