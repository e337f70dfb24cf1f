// K4-K6: the non-GEMM pieces of the MLP student distillation step (gfx950).
//   K4  log_softmax + NLL / KL(log-target, batchmean) forward AND gradient wrt logits in one pass
//       (reference train_and_eval.py:77-84 with the criteria of train_student.py:278-279)
//   K5  BatchNorm1d training statistics / backward (reference models.py:28-31,48-49)
//   K6  fused multi-tensor Adam, torch.optim.Adam semantics (reference train_student.py:275-277)
// All reductions are tree/fixed-order (no float atomics): results are run-to-run deterministic.
#include <cmath>
#include <cstdlib>

#include "glnn_common.h"
#include "student_dev.h"

namespace {

// ------------------------------------------------------------------------------------------
// K4: one wavefront per row (c <= 64: one class per lane; larger c loops), 4 rows per workgroup.
// ------------------------------------------------------------------------------------------

// LPR = 8: c <= 8 (the two-class sets pokec / penn94: 1.6 M rows) -- eight rows per wavefront, eight lanes per row, the same arithmetic per
// row; a wave per row left 62 lanes idle and took 424 us for pokec's full-graph loss.
template <int LPR>
__device__ __forceinline__ float grp_max(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
template <int LPR>
__device__ __forceinline__ float grp_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <bool LOSS, int LPR = 64>
__global__ __launch_bounds__(256) void softmax_loss_kernel(const LossArgs a) {
  constexpr int RPW = 64 / LPR;                       // rows per wavefront
  const int lane = threadIdx.x & (LPR - 1), wave = (threadIdx.x >> 6) * RPW + ((threadIdx.x & 63) / LPR);     // "wave" = row slot of the workgroup
  float wave_loss = 0.f, col_acc = 0.f;
  for (int64_t row = (int64_t)blockIdx.x * 4 * RPW + wave; row < a.rows; row += (int64_t)gridDim.x * 4 * RPW) {
    const float* zr = a.z + row * a.ldz;
    // the row's target / label is requested FIRST: behind the z_store below the compiler may not move these loads up (possible alias),
    // and index -> row -> value would be two more dependent round trips at the end of the row's chain
    const float* tr = nullptr;
    float tj0 = 0.f;
    int64_t yrow = 0;
    if (LOSS) {
      if (a.kind == GLNN_LOSS_NLL) {
        yrow = a.labels[a.label_rows ? a.label_rows[row] : row];
      } else {
        tr = a.t + (a.t_rows ? a.t_rows[row] : row) * a.ldt;
        tj0 = lane < a.c ? tr[lane] : 0.f;
      }
    }
    float zv = 0.f;
    if (a.nslab > 0 && lane < a.c) {         // fold the split-K partials of this row first (c <= 64: one class per lane, kept in zv)
#pragma unroll 8
      for (int s = 0; s < a.nslab; ++s) zv += a.slabs[(int64_t)s * a.slab_stride + row * a.c + lane];
      zv += a.bias ? a.bias[lane] : 0.f;
      a.z_store[row * a.ldz + lane] = zv;
    }
    auto Z = [&](int j) { return a.nslab > 0 ? zv : zr[j]; };
    float mx = -INFINITY;
    for (int j = lane; j < a.c; j += LPR) mx = fmaxf(mx, Z(j));
    mx = grp_max<LPR>(mx);
    float se = 0.f;
    for (int j = lane; j < a.c; j += LPR) se += expf(Z(j) - mx);
    se = grp_sum<LPR>(se);
    const float lse = mx + logf(se);
    if (!LOSS) {
      for (int j = lane; j < a.c; j += LPR) a.logp[row * a.ldl + j] = Z(j) - lse;
      continue;
    }
    float row_loss = 0.f;
    if (a.kind == GLNN_LOSS_NLL) {
      const int64_t y = yrow;
      for (int j = lane; j < a.c; j += LPR) {
        const float lp = Z(j) - lse;
        if (a.logp) a.logp[row * a.ldl + j] = lp;
        const float sm = expf(lp);
        const float g = (sm - (j == y ? 1.f : 0.f)) * a.scale;
        a.dz[row * a.ldg + j] = g;
        col_acc += g;
        if (j == y) row_loss = -lp;
      }
      row_loss = grp_sum<LPR>(row_loss);
    } else {
      float set = 0.f;
      for (int j = lane; j < a.c; j += LPR) {
        const float tj = j == lane ? tj0 : tr[j], et = expf(tj);
        set += et;
        row_loss += et * (tj - (Z(j) - lse));
      }
      set = grp_sum<LPR>(set);
      row_loss = grp_sum<LPR>(row_loss);
      for (int j = lane; j < a.c; j += LPR) {
        const float lp = Z(j) - lse;
        if (a.logp) a.logp[row * a.ldl + j] = lp;
        const float g = (expf(lp) * set - expf(j == lane ? tj0 : tr[j])) * a.scale;
        a.dz[row * a.ldg + j] = g;
        col_acc += g;
      }
    }
    wave_loss += row_loss;
  }
  if (LOSS) {
    __shared__ float s[4];
    __shared__ float sc[4][64];
    if (LPR < 64) {                                   // fold the wave's row slots first: losses of the RPW rows, column sums per class
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) { wave_loss += __shfl_xor(wave_loss, o); col_acc += __shfl_xor(col_acc, o); }
    }
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    if (ln == 0) s[wv] = wave_loss;
    sc[wv][ln] = ln < LPR ? col_acc : 0.f;
    __syncthreads();
    if (threadIdx.x == 0) st_part(&a.partial[blockIdx.x], (s[0] + s[1]) + (s[2] + s[3]), a.counter != nullptr);
    if (!a.counter && !a.defer) return;
    if (a.col_sum && threadIdx.x < 64) st_part(&a.col_partial[(int64_t)blockIdx.x * 64 + threadIdx.x], (sc[0][threadIdx.x] + sc[1][threadIdx.x]) + (sc[2][threadIdx.x] + sc[3][threadIdx.x]), a.counter != nullptr);
    if (a.defer) return;
    if (!last_workgroup(a.counter, (int)gridDim.x)) return;
    loss_fold_last(a, (int)gridDim.x, sc);
  }
}

__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ partial, int n, float inv_rows,
                                                            float* loss_out, float* loss_accum) {
  __shared__ float s[256];
  float v = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) v += partial[i];
  s[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float l = s[0] * inv_rows;
    if (loss_out) loss_out[0] = l;
    if (loss_accum) loss_accum[0] += l;
  }
}

// ------------------------------------------------------------------------------------------
// K5: BatchNorm statistics.  Stage 1: per (row-chunk, column) mean and M2 by a two-pass over the
// chunk (second pass hits L1/L2); stage 2: Chan's pairwise combine in double, fixed order.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_stats_stage2(const BnFinArgs a) { bn_finalize_columns<false>(a, blockIdx.x); }

// counters != NULL: the last row-chunk workgroup of a column block runs the stage-2 combine for its 64 columns itself.
// NS > 0: z is not in memory yet -- it is the sum of `nslab` <= NS split-K partial slabs (slabs[s][rows][h], gemm_split_partials) plus
// the bias; this kernel folds them (s ascending), stores z and takes the statistics from the registers (the GEMM's fold launch is gone).
struct SlabSrc { const float* slabs; int nslab; int64_t stride; const float* bias; float* z_out; };
template <int NS>
__global__ __launch_bounds__(256) void bn_stats_stage1(const float* __restrict__ z, int64_t ldz, int64_t rows, int h,
                                                        float* __restrict__ ws_mean, float* __restrict__ ws_m2, const BnFinArgs fin,
                                                        int* counters, const SlabSrc src) {
  const int lc = threadIdx.x & 63;
  const int col = blockIdx.x * 64 + lc;
  const int colc = col < h ? col : h - 1;
  const int rl = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.y * kBnRows;
  int64_t r1 = r0 + kBnRows;
  if (r1 > rows) r1 = rows;
  const int cnt = (int)(r1 - r0);
  __shared__ float sh[kRowLanes][64];
  float v[kRowsPerLane];
  float s = 0.f;
#pragma unroll
  for (int i0 = 0; i0 < kRowsPerLane; i0 += kUnroll) {
    if (NS == 0) {
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int64_t r = r0 + rl + 4 * (i0 + u);
        v[i0 + u] = z[(r < r1 ? r : r0) * ldz + colc];
      }
    } else {
      float part[NS > 0 ? NS : 1][kUnroll];
#pragma unroll
      for (int t = 0; t < NS; ++t) {
        const int64_t off = (int64_t)(t < src.nslab ? t : src.nslab - 1) * src.stride;      // slabs past nslab: re-read, weight 0
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
          const int64_t r = r0 + rl + 4 * (i0 + u);
          part[t][u] = src.slabs[off + (r < r1 ? r : r0) * h + colc];
        }
      }
      const float bb = src.bias ? src.bias[colc] : 0.f;
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        float acc = part[0][u];
#pragma unroll
        for (int t = 1; t < NS; ++t) acc += t < src.nslab ? part[t][u] : 0.f;
        acc += bb;
        v[i0 + u] = acc;
        const int64_t r = r0 + rl + 4 * (i0 + u);
        if (r < r1 && col < h) src.z_out[r * ldz + col] = acc;
      }
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) s += (r0 + rl + 4 * (i0 + u) < r1) ? v[i0 + u] : 0.f;
  }
  sh[rl][lc] = s;
  __syncthreads();
  const float mean = ((sh[0][lc] + sh[1][lc]) + (sh[2][lc] + sh[3][lc])) / (float)cnt;
  __syncthreads();
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < kRowsPerLane; ++i) {
    const float d = v[i] - mean;
    q = (r0 + rl + 4 * i < r1) ? fmaf(d, d, q) : q;
  }
  sh[rl][lc] = q;
  __syncthreads();
  if (rl == 0 && col < h) {
    st_part(&ws_mean[(int64_t)blockIdx.y * h + col], mean, counters != nullptr);
    st_part(&ws_m2[(int64_t)blockIdx.y * h + col], (sh[0][lc] + sh[1][lc]) + (sh[2][lc] + sh[3][lc]), counters != nullptr);
  }
  if (counters && last_workgroup(&counters[blockIdx.x], (int)gridDim.y)) bn_finalize_columns<true>(fin, blockIdx.x);
}

// BN / ReLU / dropout backward in three launches (was five + two for the bias gradient):
//   partial : per (row-chunk, column)  s1 = sum dy,  s2 = sum dy*xhat,   dy = drop'(da) * [z*a_scale+a_shift > 0]
//   apply   : every workgroup re-reduces the nchunks partials of its 64 columns (fixed order), writes
//             dz = gamma*rstd*(dy - S1/B - xhat*S2/B)  and the per-chunk column sums of dz (the bias gradient of
//             the Linear in front of the BatchNorm); chunk 0 also stores dgamma = S2, dbeta = S1
//   finalize: dbias[col] = sum over chunks (fixed order)
struct BnBwdArgs {
  const float* da; int64_t ldda; const float* z; int64_t ldz; int64_t rows; int h;
  const float* gamma; const float* mean; const float* rstd; const float* a_scale; const float* a_shift;
  uint32_t dthr; uint32_t dseed; float dscale;
  float* dz; int64_t lddz; float* dgamma; float* dbeta;
  float* ws1; float* ws2; float* ws3;   // [nchunks][h] each; ws3 may be NULL (no bias gradient wanted)
  int nchunks;
  // what bn_bwd_apply sums for S1/S2: nparts partials `pstride` floats apart starting at p1/p2 (local chunks: = ws1/ws2,
  // nchunks, h; batch split over ranks: the gathered per-rank sums).  dgamma/dbeta = the LOCAL sums: all partials
  // (local_part < 0) or partial `local_part` only.  rows_total: device float holding the global row count, or NULL.
  const float* p1; const float* p2; int nparts; int64_t pstride; int local_part; const float* rows_total;
  int* counters; float* dz_col_sum;      // counters != NULL: the last row-chunk workgroup of a column block folds ws3 into dz_col_sum
  int relu;                              // 1: the ReLU sits behind the norm (MLP / SAGE tails); 0: no ReLU in this tail (GCN: norm -> dropout)
  int defer_colsum;                      // bn_bwd_fused: ws3 (per-chunk column sums of dz) is left for the fused Adam launch to fold
  int nslab; int64_t slab_stride;        // bn_bwd_fused<NS > 0>: da = sum of nslab <= NS split-K slabs da[s * slab_stride + r * ldda + col]
  // bn_bwd_*_sk: da is never materialised -- da = dl[rows, kk] . w[kk, h] (the input gradient of a NARROW layer, kk <= 64 classes),
  // recomputed on the matrix cores by both passes
  const float* dl; int64_t lddl; const float* w; int64_t ldw; int kk; float* dw_ws; float* db_ws;
};

template <bool BN>
__device__ __forceinline__ float bn_dy(const BnBwdArgs& a, float zz, float dav, int64_t r, int col, float sc, float sf) {
  if (a.dthr) dav = glnn::drop_keep(a.dseed, a.dthr, (uint32_t)r, (uint32_t)col) ? dav * a.dscale : 0.f;
  return (!a.relu || (BN ? fmaf(zz, sc, sf) : zz) > 0.f) ? dav : 0.f;
}
// the same with the dropout decision made at launch (two instantiations): no branch per element in the streaming kernels
template <bool BN, bool DROP>
__device__ __forceinline__ float bn_dy_s(const BnBwdArgs& a, float zz, float dav, int64_t r, int col, float sc, float sf) {
  if (DROP) dav = glnn::drop_keep(a.dseed, a.dthr, (uint32_t)r, (uint32_t)col) ? dav * a.dscale : 0.f;
  return (!a.relu || (BN ? fmaf(zz, sc, sf) : zz) > 0.f) ? dav : 0.f;
}

template <bool DROP>
__global__ __launch_bounds__(256) void bn_bwd_partial(const BnBwdArgs a) {
  const int lc = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lc;
  const int colc = col < a.h ? col : a.h - 1;
  const int64_t r0 = (int64_t)blockIdx.y * kBnRows;
  int64_t r1 = r0 + kBnRows;
  if (r1 > a.rows) r1 = a.rows;
  const float mu = a.mean[colc], rs = a.rstd[colc], sc = a.a_scale[colc], sf = a.a_shift[colc];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i0 = 0; i0 < kRowsPerLane; i0 += kUnroll) {
    float zz[kUnroll], dd[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t r = r0 + rl + 4 * (i0 + u);
      const int64_t rc = r < r1 ? r : r0;
      zz[u] = a.z[rc * a.ldz + colc];
      dd[u] = a.da[rc * a.ldda + colc];
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t r = r0 + rl + 4 * (i0 + u);
      const float dyv = bn_dy_s<true, DROP>(a, zz[u], dd[u], r, colc, sc, sf);
      const float dy = r < r1 ? dyv : 0.f;
      s1 += dy;
      s2 = fmaf(dy, (zz[u] - mu) * rs, s2);
    }
  }
  __shared__ float sh1[kRowLanes][64], sh2[kRowLanes][64];
  sh1[rl][lc] = s1;
  sh2[rl][lc] = s2;
  __syncthreads();
  if (rl == 0 && col < a.h) {
    a.ws1[(int64_t)blockIdx.y * a.h + col] = (sh1[0][lc] + sh1[1][lc]) + (sh1[2][lc] + sh1[3][lc]);
    a.ws2[(int64_t)blockIdx.y * a.h + col] = (sh2[0][lc] + sh2[1][lc]) + (sh2[2][lc] + sh2[3][lc]);
  }
}

template <bool BN, bool DROP>
__global__ __launch_bounds__(256) void bn_bwd_apply(const BnBwdArgs a) {
  const int lc = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lc;
  const int colc = col < a.h ? col : a.h - 1;
  const int64_t r0 = (int64_t)blockIdx.y * kBnRows;
  int64_t r1 = r0 + kBnRows;
  if (r1 > a.rows) r1 = a.rows;
  __shared__ float sh1[kRowLanes][64], sh2[kRowLanes][64];
  float S1 = 0.f, S2 = 0.f, mu = 0.f, rs = 1.f, sc = 1.f, sf = 0.f, g = 1.f;
  if (BN) {
    // every row lane sums the same partials in the same order -> identical S1/S2 in all four lanes, no LDS hop
    // (the loads of 16 partials are issued together -- one memory round trip per batch, not per partial: the plain loop compiled to
    //  load, wait, add, branch, 32 dependent round trips at the head of every workgroup -- and added in the same ascending order)
    for (int k0 = 0; k0 < a.nparts; k0 += 16) {
      float t1[16], t2[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int k = k0 + u < a.nparts ? k0 + u : a.nparts - 1;
        t1[u] = a.p1[(int64_t)k * a.pstride + colc];
        t2[u] = a.p2[(int64_t)k * a.pstride + colc];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (k0 + u < a.nparts) { S1 += t1[u]; S2 += t2[u]; }
    }
    mu = a.mean[colc]; rs = a.rstd[colc]; sc = a.a_scale[colc]; sf = a.a_shift[colc]; g = a.gamma[colc];
    if (blockIdx.y == 0 && rl == 0 && col < a.h) {
      a.dbeta[col] = a.local_part < 0 ? S1 : a.p1[(int64_t)a.local_part * a.pstride + col];
      a.dgamma[col] = a.local_part < 0 ? S2 : a.p2[(int64_t)a.local_part * a.pstride + col];
    }
  }
  const float inv_b = 1.0f / (a.rows_total ? a.rows_total[0] : (float)a.rows);
  const float c1 = S1 * inv_b, c2 = S2 * inv_b, grs = g * rs;
  float sdz = 0.f;
#pragma unroll
  for (int i0 = 0; i0 < kRowsPerLane; i0 += kUnroll) {
    float zz[kUnroll], dd[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t r = r0 + rl + 4 * (i0 + u);
      const int64_t rc = r < r1 ? r : r0;
      zz[u] = a.z[rc * a.ldz + colc];
      dd[u] = a.da[rc * a.ldda + colc];
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t r = r0 + rl + 4 * (i0 + u);
      const float dy = bn_dy_s<BN, DROP>(a, zz[u], dd[u], r, colc, sc, sf);       // (rows past r1: the clamped row's values, discarded)
      const float out = BN ? glnn::bn_dz(grs, dy, c1, zz[u], mu, rs, c2) : dy;
      if (r < r1 && col < a.h) a.dz[r * a.lddz + col] = out;
      sdz += r < r1 ? out : 0.f;
    }
  }
  if (a.ws3) {
    sh1[rl][lc] = sdz;
    __syncthreads();
    if (rl == 0 && col < a.h) st_part(&a.ws3[(int64_t)blockIdx.y * a.h + col], (sh1[0][lc] + sh1[1][lc]) + (sh1[2][lc] + sh1[3][lc]), a.counters != nullptr);
    if (a.counters && last_workgroup(&a.counters[blockIdx.x], (int)gridDim.y)) {
      float s = 0.f;                                    // lane-split fold of the per-chunk sums, fixed order
#pragma unroll 8
      for (int k = rl; k < a.nchunks; k += kRowLanes) s += ld_part(&a.ws3[(int64_t)k * a.h + colc]);
      sh2[rl][lc] = s;
      __syncthreads();
      if (rl == 0 && col < a.h) a.dz_col_sum[col] = (sh2[0][lc] + sh2[1][lc]) + (sh2[2][lc] + sh2[3][lc]);
    }
  }
  (void)sh2;
}

// ---- the same two passes when the layer behind is NARROW (the classifier: 47 / 40 / 7 outputs) and the batch is large -------------
// da = dl . w has a reduction of <= 64: 0.8 GFLOP for MLP3w8's [4096, 2048] -- 5 us of MFMA time -- against 34 MB written by an input
// gradient GEMM and read back twice.  Both passes therefore RECOMPUTE their 128 x 64 tile of da on the matrix cores and da never exists
// in memory.  A workgroup stages its dl rows [128][2 KH] and its w panel [2 KH][64] in LDS once (coalesced float4 loads, zero behind
// kk; dl is 0.8 MB, w 0.4 MB: L2 hits), requests its z values, and each wave multiplies 32 rows x two 32-column blocks with operands
// read from LDS just in time (the register file holds the accumulators and z, not the operands: 4 waves per SIMD, the whole grid of
// MLP3w8 resident at once).  The reduction index is split between the two lane halves of v_mfma_f32_32x32x2_f32 as [0, KH) / [KH, 2 KH)
// (any pairing of k is a valid order), so a lane's A operands are KH consecutive floats of one row: ds_read_b128.
typedef float sk_f32x16 __attribute__((ext_vector_type(16)));
#ifndef GLNN_SK_WG_WAVES
#define GLNN_SK_WG_WAVES 3
#endif
#ifndef GLNN_SK_APPLY_WAVES
#define GLNN_SK_APPLY_WAVES 3
#endif

template <int KH>
struct SkLds {
  static constexpr int LDA = 2 * KH + 4;     // (2 KH + 4) * row mod 32 banks: the eight rows of one ds_read_b128 pass never collide
  float a[kBnRows][LDA];
  float b[2 * KH][64];
};

// stage operands (all 256 threads): every global load is issued before the first LDS store; the caller syncs before sk_mma
template <int KH>
__device__ __forceinline__ void sk_stage(const BnBwdArgs& g, SkLds<KH>& L, int64_t r0, int c0) {
  constexpr int K2 = 2 * KH, A4 = kBnRows * (K2 / 4), B4 = K2 * 16;
  constexpr int NA = (A4 + 255) / 256, NB = (B4 + 255) / 256;
  const int tid = threadIdx.x;
  float4 va[NA], vb[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int e = tid + 256 * i < A4 ? tid + 256 * i : A4 - 1;
    const int row = e / (K2 / 4), k4 = (e % (K2 / 4)) * 4;
    const int64_t r = r0 + row < g.rows ? r0 + row : g.rows - 1;
    const int64_t kc = k4 + 4 <= g.lddl ? k4 : g.lddl - 4;       // stay inside the row's pitch; everything at k >= kk is zeroed below
    va[i] = *reinterpret_cast<const float4*>(g.dl + r * g.lddl + kc);
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int e = tid + 256 * i < B4 ? tid + 256 * i : B4 - 1;
    const int k = e / 16, c4 = (e % 16) * 4;
    const float* p = g.w + (int64_t)(k < g.kk ? k : g.kk - 1) * g.ldw;
    vb[i] = *reinterpret_cast<const float4*>(p + (c0 + c4 + 4 <= g.h ? c0 + c4 : g.h - 4));      // h % 4 == 0; columns >= h are never used
  }
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int e = tid + 256 * i;
    const int row = e / (K2 / 4), k4 = (e % (K2 / 4)) * 4;
    float4 v = va[i];
    if (k4 + 4 > g.lddl) v = make_float4(0.f, 0.f, 0.f, 0.f);   // k4 >= lddl - 3 >= kk - 3 and k4 % 4 == 0 == lddl % 4  =>  k4 >= lddl >= kk
    v.x = k4 < g.kk ? v.x : 0.f; v.y = k4 + 1 < g.kk ? v.y : 0.f; v.z = k4 + 2 < g.kk ? v.z : 0.f; v.w = k4 + 3 < g.kk ? v.w : 0.f;
    if (e < A4) *reinterpret_cast<float4*>(&L.a[row][k4]) = v;
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int e = tid + 256 * i;
    const int k = e / 16, c4 = (e % 16) * 4;
    if (e < B4) *reinterpret_cast<float4*>(&L.b[k][c4]) = k < g.kk ? vb[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// C fragment: register i of lane (li, kk) is row (i & 3) + 8 (i >> 2) + 4 kk of the wave's 32, column li of the block
__device__ __forceinline__ void sk_load_z(const BnBwdArgs& g, int64_t wr0, int c0, int li, int kk, float (&zz)[2][16]) {
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int colc = c0 + 32 * b + li < g.h ? c0 + 32 * b + li : g.h - 1;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int64_t r = wr0 + (i & 3) + 8 * (i >> 2) + 4 * kk;
      zz[b][i] = g.z[(r < g.rows ? r : g.rows - 1) * g.ldz + colc];
    }
  }
}

template <int KH>
__device__ __forceinline__ void sk_mma(const SkLds<KH>& L, int wave, int li, int kk, sk_f32x16 (&acc)[2]) {
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
#pragma unroll
  for (int s = 0; s < KH; s += 4) {
    const float4 a4 = *reinterpret_cast<const float4*>(&L.a[32 * wave + li][kk * KH + s]);
    const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], L.b[kk * KH + s + t][32 * b + li], acc[b], 0, 0, 0);
    }
  }
}

// dy of one element, branch-free (bn_dy's arithmetic: the same hash, the same comparisons).  hcol = seed ^ ((col >> 1) * 0x85EBCA77 +
// 0x632BE5AB) is hoisted per column, `row` is the global row of the element.
template <bool DROP>
__device__ __forceinline__ float sk_dy(const BnBwdArgs& a, float dav, float zz, uint32_t hcol, bool hi, uint32_t row, float sc, float sf, bool valid) {
  if (DROP) {
    uint32_t h = hcol ^ (row * 0x9E3779B1u);
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    dav = (hi ? (h >> 16) : (h & 0xFFFFu)) >= a.dthr ? dav * a.dscale : 0.f;
  }
  const bool on = !a.relu || fmaf(zz, sc, sf) > 0.f;
  return (on && valid) ? dav : 0.f;
}

// WG: also leave a1 = dropout(relu(z * a_scale + a_shift)) of every element -- the operand of the classifier's weight gradient, in the
// arithmetic of the forward operand transform (gemm.hip XF == 2) -- zero outside the tile's valid rows / columns
template <bool DROP, bool FULL, bool WG>
__device__ __forceinline__ void sk_partial_tail(const BnBwdArgs& a, const sk_f32x16 (&acc)[2], const float (&zz)[2][16], int64_t wr0, int c0, int li,
                                                int kk, int wave, float* sh1, float* sh2, float (&a1)[2][16]) {
  const uint32_t rbase = (uint32_t)wr0 + 4u * kk;
  const int left = (int)((a.rows - wr0 < 32 ? a.rows - wr0 : 32)) - 4 * kk;     // local row (i & 3) + 8 (i >> 2) is valid iff < left
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int col = c0 + 32 * b + li;
    const int colc = FULL || col < a.h ? col : a.h - 1;
    const float mu = a.mean[colc], rs = a.rstd[colc], sc = a.a_scale[colc], sf = a.a_shift[colc];
    const uint32_t hcol = a.dseed ^ (((uint32_t)colc >> 1) * 0x85EBCA77u + 0x632BE5ABu);
    const bool hi = colc & 1;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int lr = (i & 3) + 8 * (i >> 2);
      const bool valid = FULL || lr < left;
      if (WG) {
        bool keep = true;
        if (DROP) {
          uint32_t h = hcol ^ ((rbase + lr) * 0x9E3779B1u);
          h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
          keep = (hi ? (h >> 16) : (h & 0xFFFFu)) >= a.dthr;
        }
        const float y = fmaf(zz[b][i], sc, sf);
        const bool on = y > 0.f;                                       // WG requires a.relu
        const float dy = (keep && on && valid) ? (DROP ? acc[b][i] * a.dscale : acc[b][i]) : 0.f;
        a1[b][i] = (keep && on && valid && (FULL || col < a.h)) ? (DROP ? y * a.dscale : y) : 0.f;
        s1 += dy;
        s2 = fmaf(dy, (zz[b][i] - mu) * rs, s2);
      } else {
        const float dy = sk_dy<DROP>(a, acc[b][i], zz[b][i], hcol, hi, rbase + lr, sc, sf, valid);
        s1 += dy;
        s2 = fmaf(dy, (zz[b][i] - mu) * rs, s2);
      }
    }
    sh1[(wave * 2 + kk) * 64 + 32 * b + li] = s1;
    sh2[(wave * 2 + kk) * 64 + 32 * b + li] = s2;
  }
}

template <bool DROP, bool FULL>
__device__ __forceinline__ void sk_apply_tail(const BnBwdArgs& a, const sk_f32x16 (&acc)[2], const float (&zz)[2][16], const float (&S1v)[2],
                                              const float (&S2v)[2], int64_t wr0, int c0, int li, int kk, int wave, float (&sh1)[8][64]) {
  const float inv_b = 1.0f / (a.rows_total ? a.rows_total[0] : (float)a.rows);
  const uint32_t rbase = (uint32_t)wr0 + 4u * kk;
  const int left = (int)((a.rows - wr0 < 32 ? a.rows - wr0 : 32)) - 4 * kk;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int col = c0 + 32 * b + li;
    const int colc = FULL || col < a.h ? col : a.h - 1;
    const float S1 = S1v[b], S2 = S2v[b];
    const float mu = a.mean[colc], rs = a.rstd[colc], sc = a.a_scale[colc], sf = a.a_shift[colc], grs = a.gamma[colc] * rs;
    if (blockIdx.y == 0 && wave == 0 && kk == 0 && col < a.h) {
      a.dbeta[col] = a.local_part < 0 ? S1 : a.p1[(int64_t)a.local_part * a.pstride + col];
      a.dgamma[col] = a.local_part < 0 ? S2 : a.p2[(int64_t)a.local_part * a.pstride + col];
    }
    const float c1 = S1 * inv_b, c2 = S2 * inv_b;
    const uint32_t hcol = a.dseed ^ (((uint32_t)colc >> 1) * 0x85EBCA77u + 0x632BE5ABu);
    const bool hi = colc & 1;
    float* outp = a.dz + (wr0 + 4 * kk) * a.lddz + colc;
    float sdz = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int lr = (i & 3) + 8 * (i >> 2);
      const bool valid = FULL || lr < left;
      const float dy = sk_dy<DROP>(a, acc[b][i], zz[b][i], hcol, hi, rbase + lr, sc, sf, valid);
      const float out = glnn::bn_dz(grs, dy, c1, zz[b][i], mu, rs, c2);
      if (FULL || (valid && col < a.h)) outp[(int64_t)lr * a.lddz] = out;
      sdz += valid ? out : 0.f;
    }
    sh1[wave * 2 + kk][32 * b + li] = sdz;
  }
}

template <int KH, bool DROP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KH <= 24 ? 4 : 3))) void bn_bwd_partial_sk(const BnBwdArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, kk = lane >> 5;
  const int c0 = blockIdx.x * 64;
  const int64_t r0 = (int64_t)blockIdx.y * kBnRows, wr0 = r0 + 32 * wave;
  __shared__ __attribute__((aligned(16))) SkLds<KH> L;
  float (&sh1)[8][64] = *reinterpret_cast<float (*)[8][64]>(&L.a[0][0]);        // the operand tiles are dead behind the MFMAs:
  float (&sh2)[8][64] = *reinterpret_cast<float (*)[8][64]>(&L.a[0][0] + 512);   // 4 KB of them carry the wave sums (4 workgroups per CU)
  static_assert(sizeof(L.a) >= 2 * 8 * 64 * sizeof(float), "SkLds::a too small for the wave sums");
  float zz[2][16];
  sk_load_z(a, wr0, c0, li, kk, zz);
  sk_stage<KH>(a, L, r0, c0);
  __syncthreads();
  sk_f32x16 acc[2];
  sk_mma<KH>(L, wave, li, kk, acc);
  __syncthreads();
  const bool full = r0 + kBnRows <= a.rows && c0 + 64 <= a.h;          // uniform: no per-element row / column checks
  float unused[2][16];
  if (full) sk_partial_tail<DROP, true, false>(a, acc, zz, wr0, c0, li, kk, wave, &sh1[0][0], &sh2[0][0], unused);
  else sk_partial_tail<DROP, false, false>(a, acc, zz, wr0, c0, li, kk, wave, &sh1[0][0], &sh2[0][0], unused);
  __syncthreads();
  if (threadIdx.x < 64 && c0 + (int)threadIdx.x < a.h) {
    const int c = threadIdx.x;
    a.ws1[(int64_t)blockIdx.y * a.h + c0 + c] = ((sh1[0][c] + sh1[1][c]) + (sh1[2][c] + sh1[3][c])) + ((sh1[4][c] + sh1[5][c]) + (sh1[6][c] + sh1[7][c]));
    a.ws2[(int64_t)blockIdx.y * a.h + c0 + c] = ((sh2[0][c] + sh2[1][c]) + (sh2[2][c] + sh2[3][c])) + ((sh2[4][c] + sh2[5][c]) + (sh2[6][c] + sh2[7][c]));
  }
}

// The first pass PLUS the classifier's own gradients: with dl and a1 = act(z) both on chip, dW[k, c] = sum_r dl[r, k] a1[r, c] of the
// tile is 64 more MFMAs per wave -- B operands are the a1 registers themselves (C-fragment row order = the reduction order: any pairing
// of rows is valid), A operands dl[row][class] come from the LDS tile.  The four waves' 32-row partials are summed through LDS (fixed
// order) and stored as row-chunk slab dw_ws[chunk][k][h] (folded by Adam or chunk_sum_kernel: k ascending); workgroups of the first
// column block also store the chunk's column sums of dl (the bias gradient) to db_ws[chunk][k].  Replaces a gemm_tn launch that re-read z.
template <int KH, bool DROP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KH <= 24 ? GLNN_SK_WG_WAVES : 2))) void bn_bwd_partial_wg_sk(const BnBwdArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, kk = lane >> 5;
  const int c0 = blockIdx.x * 64;
  const int64_t r0 = (int64_t)blockIdx.y * kBnRows, wr0 = r0 + 32 * wave;
  constexpr int LF = sizeof(SkLds<KH>) / 4, N = LF > 9216 ? LF : 9216;      // floats: the operand tiles, later red[4][32][64] + the wave sums
  constexpr int LDA = SkLds<KH>::LDA;
  __shared__ __attribute__((aligned(16))) float raw[N];
  SkLds<KH>& L = *reinterpret_cast<SkLds<KH>*>(raw);
  float* sh1 = raw + N - 1024;                                               // behind L.a (needed until the A operands are read) and red
  float* sh2 = raw + N - 512;
  static_assert(N - 1024 >= 8192 && N - 1024 >= (int)(sizeof(L.a) / 4), "wave sums overlap red / the dl tile");
  float zz[2][16];
  sk_load_z(a, wr0, c0, li, kk, zz);
  sk_stage<KH>(a, L, r0, c0);
  __syncthreads();
  sk_f32x16 acc[2];
  sk_mma<KH>(L, wave, li, kk, acc);
  __syncthreads();                                                           // L.b is dead: the wave sums may land in it
  const bool full = r0 + kBnRows <= a.rows && c0 + 64 <= a.h;
  float a1[2][16];
  if (full) sk_partial_tail<DROP, true, true>(a, acc, zz, wr0, c0, li, kk, wave, sh1, sh2, a1);
  else sk_partial_tail<DROP, false, true>(a, acc, zz, wr0, c0, li, kk, wave, sh1, sh2, a1);
  if (blockIdx.x == 0 && a.db_ws && threadIdx.x < a.kk) {                   // bias gradient partial: column sums of the chunk's dl rows
    const int nv = (int)(a.rows - r0 < kBnRows ? a.rows - r0 : kBnRows);
    float s = 0.f;
    for (int r = 0; r < nv; ++r) s += L.a[r][threadIdx.x];
    a.db_ws[(int64_t)blockIdx.y * a.kk + threadIdx.x] = s;
  }
  const int nmb = a.kk > 32 ? 2 : 1;
  float aop[2][16];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const int cidx = 32 * mb + li < LDA ? 32 * mb + li : LDA - 1;           // classes >= kk: finite garbage into rows of D that are never stored
#pragma unroll
    for (int i = 0; i < 16; ++i) aop[mb][i] = (mb < nmb) ? L.a[32 * wave + (i & 3) + 8 * (i >> 2) + 4 * kk][cidx] : 0.f;
  }
  __syncthreads();
  if (threadIdx.x < 64 && c0 + (int)threadIdx.x < a.h) {
    const int c = threadIdx.x;
    a.ws1[(int64_t)blockIdx.y * a.h + c0 + c] = ((sh1[c] + sh1[64 + c]) + (sh1[128 + c] + sh1[192 + c])) + ((sh1[256 + c] + sh1[320 + c]) + (sh1[384 + c] + sh1[448 + c]));
    a.ws2[(int64_t)blockIdx.y * a.h + c0 + c] = ((sh2[c] + sh2[64 + c]) + (sh2[128 + c] + sh2[192 + c])) + ((sh2[256 + c] + sh2[320 + c]) + (sh2[384 + c] + sh2[448 + c]));
  }
  float* red = raw;                                                          // [4 waves][32 classes][64 columns]
  for (int mb = 0; mb < nmb; ++mb) {
    sk_f32x16 dw[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int j = 0; j < 16; ++j) dw[nb][j] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float av = mb == 0 ? aop[0][i] : aop[1][i];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) dw[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, a1[nb][i], dw[nb], 0, 0, 0);
    }
    if (mb) __syncthreads();                                                 // the previous block's sums have been read
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int j = 0; j < 16; ++j) red[wave * 2048 + ((j & 3) + 8 * (j >> 2) + 4 * kk) * 64 + 32 * nb + li] = dw[nb][j];
    __syncthreads();
    {
      const int m = threadIdx.x >> 3, c8 = (threadIdx.x & 7) * 8, cls = 32 * mb + m;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float* rp = red + m * 64 + c8 + 4 * q;
        const float4 v0 = *reinterpret_cast<const float4*>(rp), v1 = *reinterpret_cast<const float4*>(rp + 2048);
        const float4 v2 = *reinterpret_cast<const float4*>(rp + 4096), v3 = *reinterpret_cast<const float4*>(rp + 6144);
        const float4 o = make_float4((v0.x + v1.x) + (v2.x + v3.x), (v0.y + v1.y) + (v2.y + v3.y), (v0.z + v1.z) + (v2.z + v3.z),
                                     (v0.w + v1.w) + (v2.w + v3.w));
        const int col = c0 + c8 + 4 * q;
        if (cls < a.kk && col + 4 <= a.h)                                    // h % 4 == 0
          *reinterpret_cast<float4*>(a.dw_ws + ((int64_t)blockIdx.y * a.kk + cls) * a.h + col) = o;
      }
    }
  }
}

template <int KH, bool DROP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KH <= 24 ? GLNN_SK_APPLY_WAVES : 3))) void bn_bwd_apply_sk(const BnBwdArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, kk = lane >> 5;
  const int c0 = blockIdx.x * 64;
  const int64_t r0 = (int64_t)blockIdx.y * kBnRows, wr0 = r0 + 32 * wave;
  __shared__ __attribute__((aligned(16))) SkLds<KH> L;
  __shared__ float sh1[8][64];
  float zz[2][16];
  sk_load_z(a, wr0, c0, li, kk, zz);
  sk_stage<KH>(a, L, r0, c0);
  // S1 / S2 of the 64 columns, once per workgroup: wave j sums partials j*8 .. j*8+7 (+32, ...) of column c0 + lane in ascending order
  // (16 loads in flight, under the staging loads), the four wave sums are added pairwise behind the barrier -- the same order in every
  // workgroup, so all of them hold identical S1 / S2
  {
    const int colc = c0 + lane < a.h ? c0 + lane : a.h - 1;
    float q1 = 0.f, q2 = 0.f;
    for (int k0 = wave * 8; k0 < a.nparts; k0 += 32) {
      float t1[8], t2[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + u < a.nparts ? k0 + u : a.nparts - 1;
        t1[u] = a.p1[(int64_t)k * a.pstride + colc];
        t2[u] = a.p2[(int64_t)k * a.pstride + colc];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (k0 + u < a.nparts) { q1 += t1[u]; q2 += t2[u]; }
    }
    sh1[wave][lane] = q1;
    sh1[4 + wave][lane] = q2;
  }
  __syncthreads();
  float S1v[2], S2v[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    S1v[b] = (sh1[0][32 * b + li] + sh1[1][32 * b + li]) + (sh1[2][32 * b + li] + sh1[3][32 * b + li]);
    S2v[b] = (sh1[4][32 * b + li] + sh1[5][32 * b + li]) + (sh1[6][32 * b + li] + sh1[7][32 * b + li]);
  }
  sk_f32x16 acc[2];
  sk_mma<KH>(L, wave, li, kk, acc);
  __syncthreads();                                   // sh1 is reused for the column sums of dz
  const bool full = r0 + kBnRows <= a.rows && c0 + 64 <= a.h;
  if (full) sk_apply_tail<DROP, true>(a, acc, zz, S1v, S2v, wr0, c0, li, kk, wave, sh1);
  else sk_apply_tail<DROP, false>(a, acc, zz, S1v, S2v, wr0, c0, li, kk, wave, sh1);
  if (a.ws3) {
    __syncthreads();
    if (threadIdx.x < 64 && c0 + (int)threadIdx.x < a.h) {
      const int c = threadIdx.x;
      a.ws3[(int64_t)blockIdx.y * a.h + c0 + c] = ((sh1[0][c] + sh1[1][c]) + (sh1[2][c] + sh1[3][c])) + ((sh1[4][c] + sh1[5][c]) + (sh1[6][c] + sh1[7][c]));
    }
  }
}

template <int KH>
static void launch_bn_bwd_sk(bool apply, dim3 grid, hipStream_t st, const BnBwdArgs& a) {
  if (apply) {
    if (a.dthr) hipLaunchKernelGGL((bn_bwd_apply_sk<KH, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((bn_bwd_apply_sk<KH, false>), grid, dim3(256), 0, st, a);
  } else if (a.dw_ws) {
    if (a.dthr) hipLaunchKernelGGL((bn_bwd_partial_wg_sk<KH, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((bn_bwd_partial_wg_sk<KH, false>), grid, dim3(256), 0, st, a);
  } else {
    if (a.dthr) hipLaunchKernelGGL((bn_bwd_partial_sk<KH, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((bn_bwd_partial_sk<KH, false>), grid, dim3(256), 0, st, a);
  }
}
static void launch_bn_bwd_sk(bool apply, dim3 grid, hipStream_t st, const BnBwdArgs& a) {
  const int kh = (a.kk + 1) / 2;
  if (kh <= 4) launch_bn_bwd_sk<4>(apply, grid, st, a);
  else if (kh <= 8) launch_bn_bwd_sk<8>(apply, grid, st, a);
  else if (kh <= 16) launch_bn_bwd_sk<16>(apply, grid, st, a);
  else if (kh <= 24) launch_bn_bwd_sk<24>(apply, grid, st, a);
  else launch_bn_bwd_sk<32>(apply, grid, st, a);
}

// BN / ReLU / dropout backward in ONE launch, for grids small enough to be co-resident (<= 256 workgroups: the latency-bound
// B = 512 ... 4096 students, where bn_bwd_partial + bn_bwd_apply were two ~7-9 us launches over half a megabyte): every workgroup
// keeps its 128 x 64 tile of z and dy in registers, publishes its partial sums (write-through stores), bumps the column block's
// arrival counter and WAITS until all row chunks of that column block have arrived; then the same fixed-order sums and the same
// per-element arithmetic as the two-launch form follow -- bit-identical results.  The arrival counter is counters[512 + column
// block]; the fold counter of the bias gradient stays counters[column block]; the last workgroup through the fold resets both.
template <int NS>
__global__ __launch_bounds__(256) void bn_bwd_fused(const BnBwdArgs a) {
  const int lc = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lc;
  const int colc = col < a.h ? col : a.h - 1;
  const int64_t r0 = (int64_t)blockIdx.y * kBnRows;
  int64_t r1 = r0 + kBnRows;
  if (r1 > a.rows) r1 = a.rows;
  const float mu = a.mean[colc], rs = a.rstd[colc], sc = a.a_scale[colc], sf = a.a_shift[colc], g = a.gamma[colc];
  float zz[kRowsPerLane], dy[kRowsPerLane];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i0 = 0; i0 < kRowsPerLane; i0 += kUnroll) {
    float dd[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t r = r0 + rl + 4 * (i0 + u);
      const int64_t rc = r < r1 ? r : r0;
      zz[i0 + u] = a.z[rc * a.ldz + colc];
      dd[u] = a.da[rc * a.ldda + colc];
    }
    if (NS > 1) {                 // the other split-K slabs of the input gradient (those past nslab: re-read, weight 0)
      float part[NS > 1 ? NS - 1 : 1][kUnroll];
#pragma unroll
      for (int t = 1; t < NS; ++t) {
        const int64_t off = (int64_t)(t < a.nslab ? t : a.nslab - 1) * a.slab_stride;
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
          const int64_t r = r0 + rl + 4 * (i0 + u);
          part[t - 1][u] = a.da[off + (r < r1 ? r : r0) * a.ldda + colc];
        }
      }
#pragma unroll
      for (int t = 1; t < NS; ++t)
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) dd[u] += t < a.nslab ? part[t - 1][u] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t r = r0 + rl + 4 * (i0 + u);
      const float d = r < r1 ? bn_dy<true>(a, zz[i0 + u], dd[u], r, colc, sc, sf) : 0.f;
      dy[i0 + u] = d;
      s1 += d;
      s2 = fmaf(d, (zz[i0 + u] - mu) * rs, s2);
    }
  }
  __shared__ float sh1[kRowLanes][64], sh2[kRowLanes][64];
  __shared__ int s_go;
  int departed = -1;
  sh1[rl][lc] = s1;
  sh2[rl][lc] = s2;
  __syncthreads();
  if (rl == 0 && col < a.h) {
    st_part(&a.ws1[(int64_t)blockIdx.y * a.h + col], (sh1[0][lc] + sh1[1][lc]) + (sh1[2][lc] + sh1[3][lc]), true);
    st_part(&a.ws2[(int64_t)blockIdx.y * a.h + col], (sh2[0][lc] + sh2[1][lc]) + (sh2[2][lc] + sh2[3][lc]), true);
  }
  publish_drain();
  __syncthreads();                                     // the partial stores of this workgroup have been acknowledged
  if (threadIdx.x == 0) {
    int* arrive = &a.counters[512 + blockIdx.x];
    __hip_atomic_fetch_add(arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (int)gridDim.y) __builtin_amdgcn_s_sleep(1);
    // deferred column sums: no fold tail whose last workgroup could reset the arrival counter -- count DEPARTURES from the wait
    // instead (nothing to publish: no drain), requested here so that the round trip hides under the apply phase
    if (a.defer_colsum) departed = __hip_atomic_fetch_add(&a.counters[blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_go = 1;
  }
  __syncthreads();
  (void)s_go;
  // every row lane sums the same partials in the same order -> identical S1/S2 everywhere (as bn_bwd_apply)
  float S1 = 0.f, S2 = 0.f;
  for (int k0 = 0; k0 < a.nchunks; k0 += 16) {          // 32 loads in flight per batch (the plain loop: one round trip per partial), same order
    float t1[16], t2[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int k = k0 + u < a.nchunks ? k0 + u : a.nchunks - 1;
      t1[u] = ld_part(&a.ws1[(int64_t)k * a.h + colc]);
      t2[u] = ld_part(&a.ws2[(int64_t)k * a.h + colc]);
    }
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (k0 + u < a.nchunks) { S1 += t1[u]; S2 += t2[u]; }
  }
  if (blockIdx.y == 0 && rl == 0 && col < a.h) {
    a.dbeta[col] = S1;
    a.dgamma[col] = S2;
  }
  const float inv_b = 1.0f / (float)a.rows;
  const float c1 = S1 * inv_b, c2 = S2 * inv_b, grs = g * rs;
  float sdz = 0.f;
#pragma unroll
  for (int i = 0; i < kRowsPerLane; ++i) {
    const int64_t r = r0 + rl + 4 * i;
    if (r < r1) {
      const float out = glnn::bn_dz(grs, dy[i], c1, zz[i], mu, rs, c2);
      if (col < a.h) a.dz[r * a.lddz + col] = out;
      sdz += out;
    }
  }
  __syncthreads();
  sh1[rl][lc] = sdz;
  __syncthreads();
  if (rl == 0 && col < a.h) st_part(&a.ws3[(int64_t)blockIdx.y * a.h + col], (sh1[0][lc] + sh1[1][lc]) + (sh1[2][lc] + sh1[3][lc]), !a.defer_colsum);
  if (a.defer_colsum) {
    if (threadIdx.x == 0 && departed == (int)gridDim.y - 1) {       // every workgroup of the column block is past the wait: reset both
      __hip_atomic_store(&a.counters[512 + blockIdx.x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&a.counters[blockIdx.x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  if (last_workgroup(&a.counters[blockIdx.x], (int)gridDim.y)) {
    if (threadIdx.x == 0) __hip_atomic_store(&a.counters[512 + blockIdx.x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float s = 0.f;                                      // lane-split fold of the per-chunk sums, fixed order (as bn_bwd_apply)
#pragma unroll 8
    for (int k = rl; k < a.nchunks; k += kRowLanes) s += ld_part(&a.ws3[(int64_t)k * a.h + colc]);
    sh2[rl][lc] = s;
    __syncthreads();
    if (rl == 0 && col < a.h) a.dz_col_sum[col] = (sh2[0][lc] + sh2[1][lc]) + (sh2[2][lc] + sh2[3][lc]);
  }
}

__global__ void chunk_sum_kernel(const float* __restrict__ ws, int nchunks, int h, float* __restrict__ out) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= h) return;
  float s = 0.f;
#pragma unroll 8
  for (int k = 0; k < nchunks; ++k) s += ws[(int64_t)k * h + col];
  out[col] = s;
}

// out[0:h] = sum_k ws1[k] (and out[h:2h] = sum_k ws2[k] when ws2 != NULL) with the column kernels' mapping: 64 columns x 4
// partial lanes per workgroup, lane rl sums partials rl, rl+4, ... (8 independent loads in flight), the four lane sums are
// folded in fixed order through LDS.  Used when a reduction has MANY row chunks (tens of thousands of rows: the hidden
// activations of a sampled teacher batch) -- a single thread per column walking them is a chain of ~400 dependent adds.
__global__ __launch_bounds__(256) void chunk_sum_lanes_kernel(float* __restrict__ ws1, float* __restrict__ ws2, int nchunks, int h,
                                                              int slice, float* __restrict__ out) {
  // blockIdx.y = slice of `slice` consecutive chunks; out == NULL: the slice's sum goes back into its FIRST chunk's slot (an
  // in-place tree level: no other workgroup reads that slot), else into out[col] (+ out[h + col] for ws2).
  const int lc = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lc;
  const int colc = col < h ? col : h - 1;
  const int k0 = blockIdx.y * slice;
  int k1 = k0 + slice;
  if (k1 > nchunks) k1 = nchunks;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
  for (int k = k0 + rl; k < k1; k += kRowLanes) {
    s1 += ws1[(int64_t)k * h + colc];
    if (ws2) s2 += ws2[(int64_t)k * h + colc];
  }
  __shared__ float sh1[kRowLanes][64], sh2[kRowLanes][64];
  sh1[rl][lc] = s1;
  sh2[rl][lc] = s2;
  __syncthreads();
  if (rl == 0 && col < h) {
    const float t1 = (sh1[0][lc] + sh1[1][lc]) + (sh1[2][lc] + sh1[3][lc]), t2 = (sh2[0][lc] + sh2[1][lc]) + (sh2[2][lc] + sh2[3][lc]);
    if (out) {
      out[col] = t1;
      if (ws2) out[h + col] = t2;
    } else {
      ws1[(int64_t)k0 * h + col] = t1;
      if (ws2) ws2[(int64_t)k0 * h + col] = t2;
    }
  }
}

// strided view of the slice sums a tree level left behind: chunk k of the next level = slot k * slice of this one
__global__ __launch_bounds__(256) void chunk_sum_strided_kernel(const float* __restrict__ ws1, const float* __restrict__ ws2, int nslices,
                                                                int slice, int h, float* __restrict__ out) {
  const int lc = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lc;
  const int colc = col < h ? col : h - 1;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
  for (int k = rl; k < nslices; k += kRowLanes) {
    s1 += ws1[(int64_t)k * slice * h + colc];
    if (ws2) s2 += ws2[(int64_t)k * slice * h + colc];
  }
  __shared__ float sh1[kRowLanes][64], sh2[kRowLanes][64];
  sh1[rl][lc] = s1;
  sh2[rl][lc] = s2;
  __syncthreads();
  if (rl == 0 && col < h) {
    out[col] = (sh1[0][lc] + sh1[1][lc]) + (sh1[2][lc] + sh1[3][lc]);
    if (ws2) out[h + col] = (sh2[0][lc] + sh2[1][lc]) + (sh2[2][lc] + sh2[3][lc]);
  }
}

// out[0:h] = sum over the nchunks per-chunk partials of ws1 (and out[h:2h] of ws2): one lane-split pass, or -- for the
// thousands of chunks of a 500k-row activation (products teacher training: a single pass walked ~1000 partials per lane,
// 1 ms) -- an in-place tree level over slices of 64 chunks followed by the fold of the slice sums.  Fixed summation order.
static void fold_chunks(float* ws1, float* ws2, int nchunks, int h, float* out, hipStream_t st) {
  const dim3 cols((h + 63) / 64);
  constexpr int kSlice = 64;
  if (nchunks <= 4 * kSlice) {
    hipLaunchKernelGGL(chunk_sum_lanes_kernel, dim3(cols.x, 1), dim3(256), 0, st, ws1, ws2, nchunks, h, nchunks, out);
    return;
  }
  const int nslices = (nchunks + kSlice - 1) / kSlice;
  hipLaunchKernelGGL(chunk_sum_lanes_kernel, dim3(cols.x, nslices), dim3(256), 0, st, ws1, ws2, nchunks, h, kSlice, (float*)nullptr);
  hipLaunchKernelGGL(chunk_sum_strided_kernel, cols, dim3(256), 0, st, ws1, ws2, nslices, kSlice, h, out);
}

// send[0:h] = sum_k ws1[k], send[h:2h] = sum_k ws2[k]  (this rank's S1/S2, the 2h floats exchanged in the backward)
__global__ void chunk_sum2_kernel(const float* __restrict__ ws1, const float* __restrict__ ws2, int nchunks, int h,
                                  float* __restrict__ out) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= h) return;
  float s1 = 0.f, s2 = 0.f;
  for (int k = 0; k < nchunks; ++k) {
    s1 += ws1[(int64_t)k * h + col];
    s2 += ws2[(int64_t)k * h + col];
  }
  out[col] = s1;
  out[h + col] = s2;
}

// ------------------------------------------------------------------------------------------
// K6: multi-tensor Adam. grid = (chunks, tensors), float4 where the tensor allows it.
// ------------------------------------------------------------------------------------------
constexpr int kAdamSrcMax = 32;      // tensors of one launch that may carry a fold source
struct AdamSrc { const float* src; int nslab; int lanes4; int64_t stride; };
struct AdamArgs {
  float* const* p; const float* const* g; float* const* m; float* const* v; const int64_t* sizes;
  float beta1, beta2, eps, wd, step_size, bc2_sqrt;
  int nsrc;                          // tensors [0, nsrc) consult src[t]
  AdamSrc src[kAdamSrcMax];
  int num_tensors;                   // blockIdx.y == num_tensors: the loss-fold job (one workgroup), when lf.partial != NULL
  glnn::LossFoldJob lf;
};

__global__ __launch_bounds__(256) void adam_kernel(const AdamArgs a) {
  const int t = blockIdx.y;
  if (t == a.num_tensors) {          // loss = sum of the loss kernel's per-workgroup partials / rows: loss_fold_last's sums, same order
    if (blockIdx.x != 0 || !a.lf.partial) return;
    __shared__ float red[256];
    float v = 0.f;
    for (int i = threadIdx.x; i < a.lf.nblocks; i += 256) v += a.lf.partial[i];
    red[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      const float l = red[0] * a.lf.inv_rows;
      if (a.lf.loss_out) a.lf.loss_out[0] = l;
      if (a.lf.loss_accum) a.lf.loss_accum[0] += l;
    }
    return;
  }
  const int64_t n = a.sizes[t];
  float* __restrict__ p = a.p[t];
  float* g = const_cast<float*>(a.g[t]);
  float* __restrict__ m = a.m[t];
  float* __restrict__ v = a.v[t];
  const AdamSrc sr = t < a.nsrc ? a.src[t] : AdamSrc{nullptr, 0, 0, 0};
  const float omb1 = 1.f - a.beta1, omb2 = 1.f - a.beta2;
  // one element: the update itself (identical in the scalar and the float4 walk)
  // (every contraction pinned by hand: left to the compiler, the scalar walk got mul + add for the lerp and the float4 walk a packed
  //  fma -- two roundings of the same update, and the one-call step (folding, scalar) no longer equalled the two-call step bit for bit)
  auto update = [&](float gi, float pi, float& mi, float& vi) -> float {
#pragma clang fp contract(off)
    if (a.wd != 0.f) gi = fmaf(a.wd, pi, gi);
    mi = fmaf(gi - mi, omb1, mi);                                // exp_avg.lerp_(grad, 1-beta1)
    vi = fmaf(gi * gi, omb2, vi * a.beta2);                      // mul_(beta2).addcmul_(g,g,1-beta2)
    const float denom = sqrtf(vi) / a.bc2_sqrt + a.eps;
    return fmaf(-a.step_size, mi / denom, pi);
  };
  // float4 walk: a plain gradient (no partials to fold), every array 16-byte aligned, a multiple of 4 elements -- the weight matrices of
  // the wide students (MLP3w8: 4.2 M of the 4.5 M parameters): 7 streams of 16 bytes per lane instead of 4 (30 -> 2x us at 126 MB)
  const bool vec = sr.nslab == 0 && (n & 3) == 0 &&
                   (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
  if (vec) {
    const int64_t n4 = n >> 2;
    float4* __restrict__ p4 = reinterpret_cast<float4*>(p);
    const float4* __restrict__ g4 = reinterpret_cast<const float4*>(g);
    float4* __restrict__ m4 = reinterpret_cast<float4*>(m);
    float4* __restrict__ v4 = reinterpret_cast<float4*>(v);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
      // g, m, v are streamed once per step (126 MB for MLP3w8: nothing survives in L2 until the next step): non-temporal accesses,
      // 28.6 -> 27.8 us; p is read again by the next forward
      typedef float nt4 __attribute__((ext_vector_type(4)));
      const nt4 g_ = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(g4) + i);
      const nt4 m_ = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(m4) + i);
      const nt4 v_ = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(v4) + i);
      const float4 gg = make_float4(g_.x, g_.y, g_.z, g_.w);
      float4 pp = p4[i], mm = make_float4(m_.x, m_.y, m_.z, m_.w), vv = make_float4(v_.x, v_.y, v_.z, v_.w);
      pp.x = update(gg.x, pp.x, mm.x, vv.x);
      pp.y = update(gg.y, pp.y, mm.y, vv.y);
      pp.z = update(gg.z, pp.z, mm.z, vv.z);
      pp.w = update(gg.w, pp.w, mm.w, vv.w);
      { const nt4 o = {mm.x, mm.y, mm.z, mm.w}; __builtin_nontemporal_store(o, reinterpret_cast<nt4*>(m4) + i); }
      { const nt4 o = {vv.x, vv.y, vv.z, vv.w}; __builtin_nontemporal_store(o, reinterpret_cast<nt4*>(v4) + i); }
      p4[i] = pp;
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float gi;
    if (sr.nslab == 0) {
      gi = g[i];
    } else {
      // eight partials requested together (a runtime-length loop of dependent adds waited for one load at a time: the launch took
      // 13 us instead of 5); partials past nslab are not loaded
      float l[4] = {0.f, 0.f, 0.f, 0.f};
      gi = 0.f;
      for (int k0 = 0; k0 < sr.nslab; k0 += 8) {
        float part[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) part[u] = k0 + u < sr.nslab ? sr.src[(int64_t)(k0 + u) * sr.stride + i] : 0.f;
        if (!sr.lanes4) {            // split-K slabs of a weight gradient: k ascending (split_reduce_kernel's order)
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (k0 + u < sr.nslab) gi = (k0 + u == 0) ? part[u] : gi + part[u];
        } else {                     // per-chunk column sums: four interleaved lanes, (l0 + l1) + (l2 + l3) (the fold tails' order)
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (k0 + u < sr.nslab) l[u & 3] += part[u];
        }
      }
      if (sr.lanes4) gi = (l[0] + l[1]) + (l[2] + l[3]);
      g[i] = gi;                     // the gradient itself stays observable (p.grad)
    }
    float mi = m[i], vi = v[i];
    const float pn = update(gi, p[i], mi, vi);
    m[i] = mi;
    v[i] = vi;
    p[i] = pn;
  }
}

}  // namespace

int glnn::softmax_loss(const float* logits, int64_t ldz, int64_t rows, int c, int kind, const int64_t* labels,
                       const int64_t* label_rows, const float* target_logp, int64_t ldt, const int64_t* target_rows, float lamb,
                       float* dlogits, int64_t ldg, float* logprob_out, int64_t ldl, float* loss_out, float* loss_accum,
                       float* workspace, int64_t workspace_floats, void* stream, int* counter, float* col_sum, const float* slabs,
                       int nslab, const float* bias, glnn::PendingFolds* pf) {
  // deferral replaces the counter form only, and only for small batches: Adam's threads walk the partials of their element in sequence
  // (B = 4096: 256 workgroup partials behind every bias-gradient element made the step 4 % SLOWER)
  // pf: the loss scalar is always left to the Adam launch; the bias-gradient column partials only when there are <= 32 of them (one row
  // per wave, rows <= 128) -- larger batches keep the last-workgroup fold for them (counter form) or get col_sum elsewhere
  const bool pf_cols = pf && counter && c <= 64 && pf->n < glnn::kMaxGradFolds && rows <= 128;
  const bool pf_loss = pf && !pf->has_loss && (pf_cols || !counter);
  if (!pf_cols && !pf_loss) pf = nullptr;
  GLNN_REQUIRE(logits && dlogits && workspace, "glnn_softmax_loss_f32: null pointer");
  GLNN_REQUIRE(nslab == 0 || (slabs && nslab > 0 && c <= 64), "glnn_softmax_loss_f32: split-K slabs need c <= 64");
  GLNN_REQUIRE(rows >= 1 && c >= 1 && ldz >= c && ldg >= c, "glnn_softmax_loss_f32: bad sizes");
  GLNN_REQUIRE(kind == GLNN_LOSS_NLL || kind == GLNN_LOSS_KL, "glnn_softmax_loss_f32: unknown kind %d", kind);
  if (kind == GLNN_LOSS_NLL) GLNN_REQUIRE(labels, "glnn_softmax_loss_f32: NLL needs labels");
  if (kind == GLNN_LOSS_KL) GLNN_REQUIRE(target_logp && ldt >= c, "glnn_softmax_loss_f32: KL needs target_logp");
  GLNN_REQUIRE(!logprob_out || ldl >= c, "glnn_softmax_loss_f32: ldl too small");
  GLNN_REQUIRE(!col_sum || (counter && c <= 64), "glnn_softmax_loss_f32: fused column sums need the counter and c <= 64");
  const bool narrow8 = c <= 8 && rows >= 4096;         // eight rows per wavefront (softmax_loss_kernel<.., 8>)
  const int rpb = narrow8 ? 32 : 4;
  int64_t blocks = (rows + rpb - 1) / rpb;
  const int64_t cap = counter ? 256 : 1024;            // fused finalize: fewer, longer workgroups keep the last one's fold short
  if (blocks > cap) blocks = cap;
  const int64_t need = blocks * (col_sum ? 65 : 1);
  GLNN_REQUIRE(workspace_floats >= need, "glnn_softmax_loss_f32: workspace needs >= %lld floats", (long long)need);
  LossArgs a = {};
  a.z = logits; a.ldz = ldz; a.rows = rows; a.c = c; a.kind = kind; a.labels = labels; a.label_rows = label_rows;
  a.t = target_logp; a.ldt = ldt; a.t_rows = target_rows; a.scale = lamb / (float)rows;
  a.dz = dlogits; a.ldg = ldg; a.logp = logprob_out; a.ldl = ldl; a.partial = workspace;
  a.counter = pf ? nullptr : counter; a.inv_rows = 1.0f / (float)rows; a.loss_out = loss_out; a.loss_accum = loss_accum;
  a.col_sum = col_sum; a.col_partial = workspace + blocks;
  a.defer = pf ? 1 : 0;
  if (pf) {
    pf->has_loss = 1;
    pf->loss = {workspace, (int)blocks, 1.0f / (float)rows, loss_out, loss_accum};
    if (col_sum && pf_cols) pf->e[pf->n++] = {col_sum, a.col_partial, (int)blocks, 1, 64};
  }
  a.slabs = slabs; a.nslab = nslab; a.slab_stride = rows * c; a.bias = bias; a.z_store = const_cast<float*>(logits);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (narrow8) hipLaunchKernelGGL((softmax_loss_kernel<true, 8>), dim3((unsigned)blocks), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((softmax_loss_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, st, a);
  if (!counter && !pf)
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, workspace, (int)blocks, 1.0f / (float)rows, loss_out, loss_accum);
  return glnn::check_launch("glnn_softmax_loss_f32");
}

int glnn::loss_finalize(const float* partial, int n, float inv_rows, float* loss_out, float* loss_accum, void* stream) {
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), partial, n, inv_rows, loss_out, loss_accum);
  return glnn::check_launch("glnn::loss_finalize");
}

extern "C" int glnn_softmax_loss_f32(const float* logits, int64_t ldz, int64_t rows, int c, int kind,
                                     const int64_t* labels, const int64_t* label_rows, const float* target_logp,
                                     int64_t ldt, const int64_t* target_rows, float lamb, float* dlogits, int64_t ldg,
                                     float* logprob_out, int64_t ldl, float* loss_out, float* loss_accum,
                                     float* workspace, int64_t workspace_floats, void* stream) {
  return glnn::softmax_loss(logits, ldz, rows, c, kind, labels, label_rows, target_logp, ldt, target_rows, lamb, dlogits, ldg,
                            logprob_out, ldl, loss_out, loss_accum, workspace, workspace_floats, stream, nullptr, nullptr);
}

extern "C" int glnn_log_softmax_f32(const float* logits, int64_t ldz, int64_t rows, int c, float* out, int64_t ldo,
                                    void* stream) {
  GLNN_REQUIRE(logits && out, "glnn_log_softmax_f32: null pointer");
  GLNN_REQUIRE(rows >= 0 && c >= 1 && ldz >= c && ldo >= c, "glnn_log_softmax_f32: bad sizes");
  if (rows == 0) return GLNN_OK;
  const bool narrow8 = c <= 8 && rows >= 4096;         // eight rows per wavefront
  const int rpb = narrow8 ? 32 : 4;
  int64_t blocks = (rows + rpb - 1) / rpb;
  if (blocks > 8192) blocks = 8192;
  LossArgs a = {};
  a.z = logits; a.ldz = ldz; a.rows = rows; a.c = c; a.logp = out; a.ldl = ldo;
  if (narrow8) hipLaunchKernelGGL((softmax_loss_kernel<false, 8>), dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
  else hipLaunchKernelGGL((softmax_loss_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
  return glnn::check_launch("glnn_log_softmax_f32");
}

static int run_exchange(const glnn::BnGroup* g, int64_t floats, void* stream, const char* what) {
  const int rc = g->exchange(g->ctx, g->send, g->recv, floats, stream);
  if (rc != 0) return glnn::fail(GLNN_ERR_INVALID_ARG, "%s: the exchange hook returned %d", what, rc);
  return GLNN_OK;
}

int glnn::bn_stats(const float* z, int64_t ldz, int64_t rows, int h, const float* gamma, const float* beta, float eps,
                   float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean_out,
                   float* rstd_out, float* a_scale_out, float* a_shift_out, float* workspace, int64_t workspace_floats,
                   void* stream, const glnn::BnGroup* g, int* counters, const float* slabs, int nslab, const float* bias,
                   const glnn::ColStats* cs) {
  GLNN_REQUIRE(z && a_scale_out && a_shift_out && workspace, "glnn_bn_stats_f32: null pointer");
  if (cs && !(cs->done && nslab <= 0 && cs->ws == workspace)) cs = nullptr;
  if (nslab > 0 && !(slabs && nslab <= 8 && counters && !g)) return GLNN_ERR_UNSUPPORTED;       // nothing launched: fold first, call again
  GLNN_REQUIRE(rows >= 1 && h >= 1 && ldz >= h, "glnn_bn_stats_f32: bad sizes");
  const bool tiles = cs && cs->chunk_rows > 0;          // the GEMM's partials cover fixed row chunks (same layout as stage 1's)
  GLNN_REQUIRE(!tiles || cs->chunk_rows == kBnRows, "glnn_bn_stats_f32: tile partials must cover %d rows", kBnRows);
  const int nchunks = (cs && !tiles) ? cs->nparts : (int)((rows + kBnRows - 1) / kBnRows);
  const int64_t need_ws = (cs && !tiles ? 3ll : 2ll) * nchunks * h + (nchunks > 4 * kBnGroup ? 3ll * ((nchunks + kBnGroup - 1) / kBnGroup) * h : 0);
  GLNN_REQUIRE(workspace_floats >= need_ws, "glnn_bn_stats_f32: workspace needs >= %lld floats", (long long)need_ws);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  float* ws_mean = cs ? cs->ws_mean : workspace;
  float* ws_m2 = cs ? cs->ws_m2 : workspace + (int64_t)nchunks * h;
  BnFinArgs a = {};
  if (cs && !tiles) a.ws_cnt = cs->ws_cnt;               // partials with their own row counts (the row-panel kernel's workgroups)
  a.ws_mean = ws_mean; a.ws_m2 = ws_m2; a.nparts = nchunks; a.pstride = h; a.rows = rows; a.h = h; a.gamma = gamma; a.beta = beta;
  a.eps = eps; a.momentum = momentum; a.running_mean = running_mean; a.running_var = running_var; a.nbt = num_batches_tracked;
  a.mean_out = mean_out; a.rstd_out = rstd_out; a.a_scale = a_scale_out; a.a_shift = a_shift_out;
  const dim3 fgrid((h + 63) / 64);
  const SlabSrc src = {slabs, nslab, rows * (int64_t)h, bias, const_cast<float*>(z)};
  if (cs) {
    // first pass already done in the producing GEMM's epilogue
  } else if (counters && !g) {     // one launch: the last row-chunk workgroup of every column block finishes the statistics
    const dim3 sgrid((h + 63) / 64, nchunks);
    if (nslab <= 0) hipLaunchKernelGGL(bn_stats_stage1<0>, sgrid, dim3(256), 0, st, z, ldz, rows, h, ws_mean, ws_m2, a, counters, src);
    else if (nslab <= 2) hipLaunchKernelGGL(bn_stats_stage1<2>, sgrid, dim3(256), 0, st, z, ldz, rows, h, ws_mean, ws_m2, a, counters, src);
    else if (nslab <= 4) hipLaunchKernelGGL(bn_stats_stage1<4>, sgrid, dim3(256), 0, st, z, ldz, rows, h, ws_mean, ws_m2, a, counters, src);
    else hipLaunchKernelGGL(bn_stats_stage1<8>, sgrid, dim3(256), 0, st, z, ldz, rows, h, ws_mean, ws_m2, a, counters, src);
    return glnn::check_launch("glnn_bn_stats_f32");
  }
  if (!cs) hipLaunchKernelGGL(bn_stats_stage1<0>, dim3((h + 63) / 64, nchunks), dim3(256), 0, st, z, ldz, rows, h, ws_mean, ws_m2, a, (int*)nullptr, src);
  if (!g && nchunks > 4 * kBnGroup && !(cs && !tiles)) {
    // thousands of row chunks (a 500k-row activation): combine groups of kBnGroup partial triples first (Chan's combine is
    // associative), then the group triples -- a single level walked ~1000 partials per lane (0.35 ms)
    const int ngroups = (nchunks + kBnGroup - 1) / kBnGroup;
    float* grp = workspace + 2ll * nchunks * h;               // [3][ngroups][h]
    BnFinArgs e = a;
    e.emit_cnt = grp; e.emit_mean = grp + (int64_t)ngroups * h; e.emit_m2 = grp + 2ll * ngroups * h; e.group_len = kBnGroup;
    hipLaunchKernelGGL(bn_stats_stage2, dim3(fgrid.x, ngroups), dim3(256), 0, st, e);
    a.ws_cnt = e.emit_cnt; a.ws_mean = e.emit_mean; a.ws_m2 = e.emit_m2; a.nparts = ngroups; a.pstride = h;
  }
  if (g) {
    // this rank's (count, mean, M2) -> all-gather -> the same fixed-order combine over the rank triples
    BnFinArgs e = a;
    e.emit_cnt = g->send; e.emit_mean = g->send + h; e.emit_m2 = g->send + 2 * h;
    hipLaunchKernelGGL(bn_stats_stage2, fgrid, dim3(256), 0, st, e);
    const int rc = glnn::check_launch("glnn_bn_stats_f32");
    if (rc != GLNN_OK) return rc;
    const int rx = run_exchange(g, 3ll * h, stream, "glnn_bn_stats_f32");
    if (rx != GLNN_OK) return rx;
    a.ws_cnt = g->recv; a.ws_mean = g->recv + h; a.ws_m2 = g->recv + 2 * h; a.nparts = g->world; a.pstride = 3ll * h;
    a.rows_out = g->rows_out;
  }
  hipLaunchKernelGGL(bn_stats_stage2, fgrid, dim3(256), 0, st, a);
  return glnn::check_launch("glnn_bn_stats_f32");
}

// z = a W^T + bias, then the BatchNorm1d training statistics of z -- the Linear + BatchNorm head of a hidden layer (reference
// models.py:43-47 / 110-114) with the statistics' first pass taken from the product kernel's epilogue whenever the kernel that takes
// the shape can leave it (glnn::gemm_stats); otherwise exactly glnn_gemm_f32 followed by glnn_bn_stats_f32.
extern "C" int glnn_linear_bn_stats_f32(const float* a, int64_t lda, int64_t m, int k, const float* w, int64_t ldw, int n, const float* bias,
                                        float* z, int64_t ldz, const float* gamma, const float* beta, float eps, float momentum,
                                        float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean_out,
                                        float* rstd_out, float* a_scale_out, float* a_shift_out, float* ws_gemm, int64_t ws_gemm_floats,
                                        float* ws_bn, int64_t ws_bn_floats, void* stream) {
  GLNN_REQUIRE(a && w && z && ws_bn, "glnn_linear_bn_stats_f32: null pointer");
  glnn::ColStats cs = {ws_bn, ws_bn_floats, 0, 0, 0, nullptr, nullptr, nullptr};
  int rc;
  if (glnn::opts().gemm_stats) rc = glnn::gemm_stats(a, lda, m, k, w, ldw, n, bias, z, ldz, ws_gemm, ws_gemm_floats, stream, &cs);
  else rc = glnn_gemm_f32(a, lda, nullptr, nullptr, nullptr, 0.f, 0u, m, k, w, ldw, 0, n, nullptr, nullptr, bias, 0, z, ldz, ws_gemm, ws_gemm_floats, stream);
  if (rc != GLNN_OK) return rc;
  return glnn::bn_stats(z, ldz, m, n, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, mean_out, rstd_out,
                        a_scale_out, a_shift_out, ws_bn, ws_bn_floats, stream, nullptr, nullptr, nullptr, 0, nullptr, cs.done ? &cs : nullptr);
}

extern "C" int glnn_bn_stats_f32(const float* z, int64_t ldz, int64_t rows, int h, const float* gamma, const float* beta,
                                 float eps, float momentum, float* running_mean, float* running_var,
                                 int64_t* num_batches_tracked, float* mean_out, float* rstd_out, float* a_scale_out,
                                 float* a_shift_out, float* workspace, int64_t workspace_floats, void* stream) {
  return glnn::bn_stats(z, ldz, rows, h, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, mean_out,
                        rstd_out, a_scale_out, a_shift_out, workspace, workspace_floats, stream, nullptr, nullptr, nullptr, 0, nullptr);
}

// bn_bwd_fused WAITS inside an ordinary launch: every workgroup of its grid must be resident at once or the waiting ones
// starve the rest (a hang).  Limit = HALF of what the current device can hold of this kernel (occupancy x CUs, queried once
// per device: a CPX/QPX partition exposes 32/64 CUs, and concurrent kernels -- the RCCL kernels of an overlapped gradient
// exchange -- may hold CUs), never more than 256; larger grids keep the two-launch form.
static int fused_grid_limit() {
  static int limit[64];
  static bool known[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  if (!known[dev]) {
    int per_cu = 0;
    hipDeviceProp_t prop;
    int lim = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, bn_bwd_fused<8>, 256, 0) == hipSuccess &&
        hipGetDeviceProperties(&prop, dev) == hipSuccess)
      lim = per_cu * prop.multiProcessorCount / 2;
    limit[dev] = lim > 256 ? 256 : lim;
    known[dev] = true;
  }
  return limit[dev];
}

namespace {
__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ src, int64_t lds, int64_t rows, int cols, float* __restrict__ dst,
                                                       int64_t ldd) {
  const int q4 = (int)(ldd >> 2);
  const int64_t total = rows * q4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / q4;
    const int c = (int)(i - r * q4) * 4;
    const float* p = src + r * lds + c;
    float4 v;
    v.x = c < cols ? p[0] : 0.f; v.y = c + 1 < cols ? p[1] : 0.f; v.z = c + 2 < cols ? p[2] : 0.f; v.w = c + 3 < cols ? p[3] : 0.f;
    *reinterpret_cast<float4*>(dst + r * ldd + c) = v;
  }
}
}  // namespace

int glnn::pad_rows(const float* src, int64_t lds, int64_t rows, int cols, float* dst, int64_t ldd, void* stream) {
  GLNN_REQUIRE(src && dst && rows >= 1 && cols >= 1 && lds >= cols && ldd >= cols && ldd % 4 == 0 && glnn::aligned16(dst), "glnn::pad_rows: bad arguments");
  int64_t blocks = (rows * (ldd >> 2) + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, lds, rows, cols, dst, ldd);
  return glnn::check_launch("glnn::pad_rows");
}

int glnn::chunk_sum(const float* ws, int nchunks, int h, float* out, void* stream) {
  GLNN_REQUIRE(ws && out && nchunks >= 1 && h >= 1, "glnn::chunk_sum: bad arguments");
  hipLaunchKernelGGL(chunk_sum_kernel, dim3((h + 127) / 128), dim3(128), 0, reinterpret_cast<hipStream_t>(stream), ws, nchunks, h, out);
  return glnn::check_launch("glnn::chunk_sum");
}

namespace {
// ---- deferred apply (bn_relu_bwd(..., defer_apply), round 5) -------------------------------------------------------------------------
// The partial pass for a consumer that evaluates dz itself: same row -> lane assignment, same accumulation order and same fold as
// bn_bwd_partial (the per-chunk sums are its bits), but a lane owns FOUR columns (float4 rows: one wave covers 256 columns of a row in
// one request) and dy -- da behind the tail's dropout and ReLU masks -- is stored in place of da.
template <bool DROP>
__global__ __launch_bounds__(256) void bn_bwd_partial_dy4(const BnBwdArgs a, float* dy_out, int64_t lddy) {      // (dy_out aliases a.da: in place)
  auto ld4g = [](const float* p) { return *reinterpret_cast<const float4*>(p); };
  const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int col4 = (blockIdx.x * 64 + lane) * 4;
  const bool col_ok = col4 < a.h;                       // h % 4 == 0: a lane's four columns are all inside or all outside
  const int cc = col_ok ? col4 : 0;
  const int64_t r0 = (int64_t)blockIdx.y * kBnRows;
  int64_t r1 = r0 + kBnRows;
  if (r1 > a.rows) r1 = a.rows;
  const float4 mu = ld4g(a.mean + cc), rs = ld4g(a.rstd + cc), sc = ld4g(a.a_scale + cc), sf = ld4g(a.a_shift + cc);
  const float mu4[4] = {mu.x, mu.y, mu.z, mu.w}, rs4[4] = {rs.x, rs.y, rs.z, rs.w}, sc4[4] = {sc.x, sc.y, sc.z, sc.w}, sf4[4] = {sf.x, sf.y, sf.z, sf.w};
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int i0 = 0; i0 < kRowsPerLane; i0 += kUnroll) {       // 8 rows x 2 float4 per lane in flight; not unrolled further (registers -> occupancy)
    float4 zz[kUnroll], dd[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t r = r0 + rl + 4 * (i0 + u);
      const int64_t rc = r < r1 ? r : r0;
      zz[u] = ld4g(a.z + rc * a.ldz + cc);
      dd[u] = ld4g(a.da + rc * a.ldda + cc);
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t r = r0 + rl + 4 * (i0 + u);
      const float z4[4] = {zz[u].x, zz[u].y, zz[u].z, zz[u].w}, d4[4] = {dd[u].x, dd[u].y, dd[u].z, dd[u].w};
      float o[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float dyv = bn_dy_s<true, DROP>(a, z4[t], d4[t], r, cc + t, sc4[t], sf4[t]);
        const float dy = r < r1 ? dyv : 0.f;
        s1[t] += dy;
        s2[t] = fmaf(dy, (z4[t] - mu4[t]) * rs4[t], s2[t]);
        o[t] = dy;
      }
      if (r < r1 && col_ok) *reinterpret_cast<float4*>(dy_out + r * lddy + col4) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
  __shared__ float4 sh1[kRowLanes][64], sh2[kRowLanes][64];
  sh1[rl][lane] = make_float4(s1[0], s1[1], s1[2], s1[3]);
  sh2[rl][lane] = make_float4(s2[0], s2[1], s2[2], s2[3]);
  __syncthreads();
  if (rl == 0 && col_ok) {
    const float4 a0 = sh1[0][lane], a1 = sh1[1][lane], a2 = sh1[2][lane], a3 = sh1[3][lane];
    const float4 b0 = sh2[0][lane], b1 = sh2[1][lane], b2 = sh2[2][lane], b3 = sh2[3][lane];
    *reinterpret_cast<float4*>(a.ws1 + (int64_t)blockIdx.y * a.h + col4) =
        make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w));
    *reinterpret_cast<float4*>(a.ws2 + (int64_t)blockIdx.y * a.h + col4) =
        make_float4((b0.x + b1.x) + (b2.x + b3.x), (b0.y + b1.y) + (b2.y + b3.y), (b0.z + b1.z) + (b2.z + b3.z), (b0.w + b1.w) + (b2.w + b3.w));
  }
}

// the column totals S1 / S2 -> dz = alpha dy + beta z + gamma  [= g rs (dy - S1/B - (z - mu) rs S2/B)], dgamma = S2, dbeta = S1; the bias
// gradient in front of the BatchNorm -- the column sum of dz, mathematically 0 -- is written as 0
__global__ __launch_bounds__(256) void bn_bwd_consts_kernel(const float* __restrict__ totals, int h, float rows, const float* __restrict__ gamma,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            float* __restrict__ alpha, float* __restrict__ beta, float* __restrict__ gam,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dz_col_sum) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= h) return;
  const float S1 = totals[col], S2 = totals[h + col];
  const float inv_b = 1.0f / rows;
  const float c1 = S1 * inv_b, c2 = S2 * inv_b, rs = rstd[col], grs = gamma[col] * rs;
  const float k = grs * rs * c2;
  alpha[col] = grs;
  beta[col] = -k;
  gam[col] = fmaf(k, mean[col], -grs * c1);
  dbeta[col] = S1;
  dgamma[col] = S2;
  if (dz_col_sum) dz_col_sum[col] = 0.f;
}
// the same behind per-tile partials that are still unfolded (glnn::gemm_bn_dy): a thread sums its column's nparts partials (all loads of a
// batch of 16 in flight, added ascending) and goes on to the constants -- fold + constants in ONE launch
__global__ __launch_bounds__(256) void bn_bwd_parts_consts_kernel(const float* __restrict__ p1, const float* __restrict__ p2, int nparts, int h,
                                                                  float rows, const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd, float* __restrict__ alpha, float* __restrict__ beta,
                                                                  float* __restrict__ gam, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                  float* __restrict__ dz_col_sum) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= h) return;
  float S1 = 0.f, S2 = 0.f;
  for (int k0 = 0; k0 < nparts; k0 += 16) {
    float t1[16], t2[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int k = k0 + u < nparts ? k0 + u : nparts - 1;
      t1[u] = p1[(int64_t)k * h + col];
      t2[u] = p2[(int64_t)k * h + col];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (k0 + u < nparts) { S1 += t1[u]; S2 += t2[u]; }
  }
  const float inv_b = 1.0f / rows;
  const float c1 = S1 * inv_b, c2 = S2 * inv_b, rs = rstd[col], grs = gamma[col] * rs;
  const float k = grs * rs * c2;
  alpha[col] = grs;
  beta[col] = -k;
  gam[col] = fmaf(k, mean[col], -grs * c1);
  dbeta[col] = S1;
  dgamma[col] = S2;
  if (dz_col_sum) dz_col_sum[col] = 0.f;
}
}  // namespace

int glnn::bn_bwd_parts_finish(const float* s1, const float* s2, int nparts, int h, int64_t rows, const float* z, int64_t ldz, const float* gamma,
                              const float* mean, const float* rstd, float* cst, float* dgamma, float* dbeta, float* dz_col_sum,
                              glnn::BnApplyA* defer_apply, void* stream) {
  GLNN_REQUIRE(s1 && s2 && nparts >= 1 && h >= 4 && (h & 3) == 0 && rows >= 1 && gamma && mean && rstd && cst && glnn::aligned16(cst) && dgamma &&
               dbeta && defer_apply, "glnn::bn_bwd_parts_finish: bad arguments");
  hipLaunchKernelGGL(bn_bwd_parts_consts_kernel, dim3((h + 63) / 64), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), s1, s2, nparts, h, (float)rows,
                     gamma, mean, rstd, cst, cst + h, cst + 2ll * h, dgamma, dbeta, dz_col_sum);
  *defer_apply = {z, ldz, cst, cst + h, cst + 2ll * h};
  return glnn::check_launch("glnn::bn_bwd_parts_finish");
}

int glnn::bn_bwd_deferred_finish(float* ws, int64_t ws_floats, int nslots, int h, int64_t rows, const float* z, int64_t ldz, const float* gamma,
                                 const float* mean, const float* rstd, float* dgamma, float* dbeta, float* dz_col_sum, glnn::BnApplyA* defer_apply,
                                 void* stream) {
  GLNN_REQUIRE(ws && glnn::aligned16(ws) && nslots >= 1 && h >= 4 && (h & 3) == 0 && rows >= 1 && gamma && mean && rstd && dgamma && dbeta &&
               defer_apply && ws_floats >= 2ll * nslots * h + 5ll * h + 8, "glnn::bn_bwd_deferred_finish: bad arguments");
  float* totals = ws + 2ll * nslots * h;
  float* cst = totals + 2ll * h;                          // (2 nslots h + 2 h: a multiple of 4 floats behind an aligned base)
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  fold_chunks(ws, ws + (int64_t)nslots * h, nslots, h, totals, st);
  hipLaunchKernelGGL(bn_bwd_consts_kernel, dim3((h + 255) / 256), dim3(256), 0, st, totals, h, (float)rows, gamma, mean, rstd, cst, cst + h,
                     cst + 2ll * h, dgamma, dbeta, dz_col_sum);
  *defer_apply = {z, ldz, cst, cst + h, cst + 2ll * h};
  return glnn::check_launch("glnn::bn_bwd_deferred_finish");
}

int glnn::bn_relu_bwd(const float* da, int64_t ldda, const float* z, int64_t ldz, int64_t rows, int h, const float* gamma,
                      const float* mean, const float* rstd, const float* a_scale, const float* a_shift, float drop_p,
                      uint32_t drop_seed, float* dz, int64_t lddz, float* dgamma, float* dbeta, float* dz_col_sum,
                      float* workspace, int64_t workspace_floats, void* stream, const glnn::BnGroup* g, int* counters, int relu,
                      int da_slabs, glnn::GradFold* defer_colsum, const glnn::NarrowProduct* prod, glnn::BnApplyA* defer_apply) {
  if (defer_colsum) *defer_colsum = {dz_col_sum, nullptr, 0, 0, 0};
  if (defer_apply) {
    // dy in place of da + per-chunk sums + fold + constants; the apply is left to the ONE consumer of dz (gemm_tn's operand pieces)
    const int nch = (int)((rows + kBnRows - 1) / kBnRows);
    if (!gamma || g || prod || da_slabs > 1 || nch <= kManyChunks || (h & 3) || !da || da != dz || ldda != lddz || (ldda & 3) || (ldz & 3) ||
        !glnn::aligned16(da) || !glnn::aligned16(z) || !glnn::aligned16(mean) || !glnn::aligned16(rstd) || !glnn::aligned16(a_scale) ||
        !glnn::aligned16(a_shift) || !workspace || !glnn::aligned16(workspace) || workspace_floats < 2ll * nch * h + 5ll * h + 8 || !mean ||
        !rstd || !a_scale || !a_shift || !dgamma || !dbeta || drop_p < 0.f || drop_p >= 1.f)
      return GLNN_ERR_UNSUPPORTED;
    BnBwdArgs a = {};
    a.da = da; a.ldda = ldda; a.z = z; a.ldz = ldz; a.rows = rows; a.h = h; a.gamma = gamma; a.mean = mean; a.rstd = rstd;
    a.a_scale = a_scale; a.a_shift = a_shift; a.dthr = glnn::drop_threshold(drop_p); a.dseed = drop_seed; a.dscale = 1.0f / (1.0f - drop_p);
    a.nchunks = nch; a.relu = relu;
    a.ws1 = workspace; a.ws2 = workspace + (int64_t)nch * h;
    float* totals = workspace + 2ll * nch * h;
    float* cst = totals + 2ll * h + ((4 - ((2ll * nch * h + 2ll * h) & 3)) & 3);          // 16-byte aligned: float4 loads in the consumer
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((h + 255) / 256, nch);
    if (a.dthr) hipLaunchKernelGGL(bn_bwd_partial_dy4<true>, grid, dim3(256), 0, st, a, dz, lddz);
    else hipLaunchKernelGGL(bn_bwd_partial_dy4<false>, grid, dim3(256), 0, st, a, dz, lddz);
    fold_chunks(a.ws1, a.ws2, nch, h, totals, st);
    hipLaunchKernelGGL(bn_bwd_consts_kernel, dim3((h + 255) / 256), dim3(256), 0, st, totals, h, (float)rows, gamma, mean, rstd, cst, cst + h,
                       cst + 2ll * h, dgamma, dbeta, dz_col_sum);
    *defer_apply = {z, ldz, cst, cst + h, cst + 2ll * h};
    return glnn::check_launch("glnn_bn_relu_bwd_f32(deferred apply)");
  }
  if (prod) {                                            // da = dl . w, recomputed by both passes (bn_bwd_*_sk): two-launch BatchNorm form only
    if (!gamma || g || da_slabs > 1 || prod->k < 1 || prod->k > 64 || !prod->dl || !prod->w || prod->lddl < prod->k || prod->ldw < h ||
        (prod->lddl | prod->ldw | h) % 4 != 0 || !glnn::aligned16(prod->dl) || !glnn::aligned16(prod->w) ||    // float4 staging loads
        (prod->dw_ws && (!relu || !glnn::aligned16(prod->dw_ws))))
      return GLNN_ERR_UNSUPPORTED;
    counters = nullptr;
    da = prod->dl; ldda = h;                             // placeholders for the checks below; the kernels never read a.da
  }
  GLNN_REQUIRE(da && z && dz, "glnn_bn_relu_bwd_f32: null pointer");
  GLNN_REQUIRE(rows >= 1 && h >= 1 && ldda >= h && ldz >= h && lddz >= h, "glnn_bn_relu_bwd_f32: bad sizes");
  GLNN_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "glnn_bn_relu_bwd_f32: drop_p must be in [0,1)");
  const int nchunks = (int)((rows + kBnRows - 1) / kBnRows);
  const bool prereduce = gamma && !g && nchunks > kManyChunks;     // many row chunks: fold S1/S2 once instead of in every workgroup
  const int64_t need = (int64_t)nchunks * h * ((gamma ? 2 : 0) + (dz_col_sum ? 1 : 0)) + (prereduce ? 2ll * h : 0);
  GLNN_REQUIRE(need == 0 || (workspace && workspace_floats >= need), "glnn_bn_relu_bwd_f32: workspace needs >= %lld floats", (long long)need);
  BnBwdArgs a;
  a.da = da; a.ldda = ldda; a.z = z; a.ldz = ldz; a.rows = rows; a.h = h; a.gamma = gamma; a.mean = mean; a.rstd = rstd;
  a.a_scale = a_scale; a.a_shift = a_shift; a.dthr = glnn::drop_threshold(drop_p); a.dseed = drop_seed;
  a.dscale = 1.0f / (1.0f - drop_p); a.dz = dz; a.lddz = lddz; a.dgamma = dgamma; a.dbeta = dbeta; a.nchunks = nchunks;
  a.counters = (dz_col_sum && counters) ? counters : nullptr; a.dz_col_sum = dz_col_sum;
  a.relu = relu ? 1 : 0; a.defer_colsum = 0;
  a.nslab = da_slabs > 0 ? da_slabs : 1; a.slab_stride = rows * ldda;
  a.dl = nullptr; a.lddl = 0; a.w = nullptr; a.ldw = 0; a.kk = 0; a.dw_ws = nullptr; a.db_ws = nullptr;
  if (prod) {
    a.dl = prod->dl; a.lddl = prod->lddl; a.w = prod->w; a.ldw = prod->ldw; a.kk = prod->k; a.dw_ws = prod->dw_ws; a.db_ws = prod->db_ws;
  }
  float* w = workspace;
  a.ws1 = a.ws2 = a.ws3 = nullptr;
  if (gamma) { a.ws1 = w; a.ws2 = w + (int64_t)nchunks * h; w += 2ll * nchunks * h; }
  if (dz_col_sum) { a.ws3 = w; w += (int64_t)nchunks * h; }
  float* totals = prereduce ? w : nullptr;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((h + 63) / 64, nchunks);
  a.p1 = a.ws1; a.p2 = a.ws2; a.nparts = nchunks; a.pstride = h; a.local_part = -1; a.rows_total = nullptr;
  const bool one_launch = glnn::opts().bn_bwd_one_launch != 0;       // 0: keep partial + apply as two launches (A/B runs, tests)
  if (gamma && !g && !prereduce && a.counters && one_launch && (int64_t)grid.x * grid.y <= fused_grid_limit() && grid.x <= 256) {
    GLNN_REQUIRE(mean && rstd && a_scale && a_shift && dgamma && dbeta, "glnn_bn_relu_bwd_f32: BN path needs stats and outputs");
    // co-resident grid: partial -> wait -> apply in one launch
    if (defer_colsum && nchunks <= 8) {   // the workspace must then stay untouched until the fused Adam launch has run
      a.defer_colsum = 1;
      *defer_colsum = {dz_col_sum, a.ws3, nchunks, 1, (int64_t)h};
    }
    if (da_slabs <= 1) hipLaunchKernelGGL(bn_bwd_fused<1>, grid, dim3(256), 0, st, a);
    else if (da_slabs <= 2) hipLaunchKernelGGL(bn_bwd_fused<2>, grid, dim3(256), 0, st, a);
    else if (da_slabs <= 4) hipLaunchKernelGGL(bn_bwd_fused<4>, grid, dim3(256), 0, st, a);
    else if (da_slabs <= 8) hipLaunchKernelGGL(bn_bwd_fused<8>, grid, dim3(256), 0, st, a);
    else return GLNN_ERR_UNSUPPORTED;
    return glnn::check_launch("glnn_bn_relu_bwd_f32");
  }
  if (da_slabs > 1) return GLNN_ERR_UNSUPPORTED;       // nothing launched: only the one-launch form folds split-K slabs
  if (gamma) {
    GLNN_REQUIRE(mean && rstd && a_scale && a_shift && dgamma && dbeta, "glnn_bn_relu_bwd_f32: BN path needs stats and outputs");
    if (prod) launch_bn_bwd_sk(false, grid, st, a);
    else if (a.dthr) hipLaunchKernelGGL(bn_bwd_partial<true>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(bn_bwd_partial<false>, grid, dim3(256), 0, st, a);
    if (g) {
      hipLaunchKernelGGL(chunk_sum2_kernel, dim3((h + 127) / 128), dim3(128), 0, st, a.ws1, a.ws2, nchunks, h, g->send);
      const int rc = glnn::check_launch("glnn_bn_relu_bwd_f32");
      if (rc != GLNN_OK) return rc;
      const int rx = run_exchange(g, 2ll * h, stream, "glnn_bn_relu_bwd_f32");
      if (rx != GLNN_OK) return rx;
      a.p1 = g->recv; a.p2 = g->recv + h; a.nparts = g->world; a.pstride = 2ll * h; a.local_part = g->rank;
      a.rows_total = g->rows_out;
    } else if (prereduce) {
      fold_chunks(a.ws1, a.ws2, nchunks, h, totals, st);
      a.p1 = totals; a.p2 = totals + h; a.nparts = 1; a.pstride = 0;
    }
    if (prod) launch_bn_bwd_sk(true, grid, st, a);
    else if (a.dthr) hipLaunchKernelGGL((bn_bwd_apply<true, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((bn_bwd_apply<true, false>), grid, dim3(256), 0, st, a);
  } else {
    if (a.dthr) hipLaunchKernelGGL((bn_bwd_apply<false, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((bn_bwd_apply<false, false>), grid, dim3(256), 0, st, a);
  }
  if (dz_col_sum && !a.counters && defer_colsum && nchunks <= 64) {
    // two-launch form before the fused Adam launch: the per-chunk column sums are folded there (chunk_sum_kernel's order: k ascending)
    *defer_colsum = {dz_col_sum, a.ws3, nchunks, 0, (int64_t)h};
  } else if (dz_col_sum && !a.counters) {
    if (nchunks > kManyChunks)
      fold_chunks(a.ws3, nullptr, nchunks, h, dz_col_sum, st);
    else
      hipLaunchKernelGGL(chunk_sum_kernel, dim3((h + 127) / 128), dim3(128), 0, st, a.ws3, nchunks, h, dz_col_sum);
  }
  return glnn::check_launch("glnn_bn_relu_bwd_f32");
}

extern "C" int glnn_bn_relu_bwd_f32(const float* da, int64_t ldda, const float* z, int64_t ldz, int64_t rows, int h,
                                    const float* gamma, const float* mean, const float* rstd, const float* a_scale,
                                    const float* a_shift, float drop_p, uint32_t drop_seed, float* dz, int64_t lddz,
                                    float* dgamma, float* dbeta, float* dz_col_sum, float* workspace,
                                    int64_t workspace_floats, void* stream) {
  return glnn::bn_relu_bwd(da, ldda, z, ldz, rows, h, gamma, mean, rstd, a_scale, a_shift, drop_p, drop_seed, dz, lddz, dgamma,
                           dbeta, dz_col_sum, workspace, workspace_floats, stream, nullptr);
}

// the same with the tail's ReLU optional: relu = 0 is the backward of  norm -> dropout  (GCN.forward, reference models.py:189-199,
// where the ReLU sits INSIDE the GraphConv in front of the norm and is differentiated separately)
extern "C" int glnn_bn_bwd_f32(const float* da, int64_t ldda, const float* z, int64_t ldz, int64_t rows, int h, const float* gamma,
                               const float* mean, const float* rstd, const float* a_scale, const float* a_shift, int relu, float drop_p,
                               uint32_t drop_seed, float* dz, int64_t lddz, float* dgamma, float* dbeta, float* dz_col_sum,
                               float* workspace, int64_t workspace_floats, void* stream) {
  return glnn::bn_relu_bwd(da, ldda, z, ldz, rows, h, gamma, mean, rstd, a_scale, a_shift, drop_p, drop_seed, dz, lddz, dgamma,
                           dbeta, dz_col_sum, workspace, workspace_floats, stream, nullptr, nullptr, relu);
}

// y = dropout(relu(z * a_scale + a_shift)): the materialised form of the operand transform the GEMM loaders apply on the
// fly, for consumers that GATHER the activation (the next SAGE layer's aggregation over a sampled block).
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ z, int64_t ldz, int64_t rows, int h,
                                                      const float* __restrict__ a_scale, const float* __restrict__ a_shift,
                                                      uint32_t thr, uint32_t seed, float dscale, float* __restrict__ y, int64_t ldy,
                                                      bool vec_cols, bool relu) {
  constexpr int U = 4;                   // float4s in flight per thread: one per pass left the kernel at 2.9 TB/s (22 us for 2 x 32 MB)
  const int h4 = (h + 3) >> 2;
  const int64_t total = rows * h4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < total; i0 += U * stride) {
    float4 v[U];
    int64_t r[U];
    int c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * stride;
      const int64_t ic = i < total ? i : i0;
      r[u] = ic / h4;
      c[u] = (int)(ic - r[u] * h4) * 4;
      v[u] = *reinterpret_cast<const float4*>(z + r[u] * ldz + c[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (i0 + u * stride >= total) break;
      float o[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
      // per-column scale / shift: ONE 16-byte load each (eight 4-byte loads per float4 halved the kernel's rate: 2.9 vs 5.7 TB/s)
      float sc[4] = {1.f, 1.f, 1.f, 1.f}, sf[4] = {0.f, 0.f, 0.f, 0.f};
      if (a_scale) {
        if (vec_cols && c[u] + 4 <= h) {
          const float4 s4 = *reinterpret_cast<const float4*>(a_scale + c[u]), f4 = *reinterpret_cast<const float4*>(a_shift + c[u]);
          sc[0] = s4.x; sc[1] = s4.y; sc[2] = s4.z; sc[3] = s4.w;
          sf[0] = f4.x; sf[1] = f4.y; sf[2] = f4.z; sf[3] = f4.w;
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (c[u] + t < h) { sc[t] = a_scale[c[u] + t]; sf[t] = a_shift[c[u] + t]; }
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float x = 0.f;
        if (c[u] + t < h) {
          x = a_scale ? fmaf(o[t], sc[t], sf[t]) : o[t];
          if (relu) x = fmaxf(x, 0.f);
          if (thr) x = glnn::drop_keep(seed, thr, (uint32_t)r[u], (uint32_t)(c[u] + t)) ? x * dscale : 0.f;
        }
        o[t] = x;                          // padding columns are written as zero
      }
      *reinterpret_cast<float4*>(y + r[u] * ldy + c[u]) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

__global__ void dropout_mask_kernel(int64_t rows, int h, uint32_t thr, uint32_t seed, uint8_t* mask) {
  const int64_t total = rows * h;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / h;
    mask[i] = (thr == 0 || glnn::drop_keep(seed, thr, (uint32_t)r, (uint32_t)(i - r * h))) ? 1 : 0;
  }
}

static int act_fwd_impl(const float* z, int64_t ldz, int64_t rows, int h, const float* a_scale, const float* a_shift, int relu,
                        float drop_p, uint32_t drop_seed, float* y, int64_t ldy, void* stream) {
  GLNN_REQUIRE(z && y, "glnn_act_fwd_f32: null pointer");
  GLNN_REQUIRE(rows >= 0 && h >= 1, "glnn_act_fwd_f32: bad sizes");
  const int64_t hp = (h + 3) & ~3;
  GLNN_REQUIRE(ldz >= hp && ldy >= hp && ldz % 4 == 0 && ldy % 4 == 0 && glnn::aligned16(z) && glnn::aligned16(y),
               "glnn_act_fwd_f32: rows must be float4 rows (leading dimensions multiples of 4, >= round4(h), 16-byte aligned)");
  GLNN_REQUIRE((a_scale == nullptr) == (a_shift == nullptr), "glnn_act_fwd_f32: a_scale and a_shift go together");
  GLNN_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "glnn_act_fwd_f32: drop_p must be in [0,1)");
  if (rows == 0) return GLNN_OK;
  int64_t blocks = (rows * (hp / 4) + 4 * 256 - 1) / (4 * 256);      // four float4s per thread
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(act_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), z, ldz, rows, h, a_scale,
                     a_shift, glnn::drop_threshold(drop_p), drop_seed, 1.0f / (1.0f - drop_p), y, ldy,
                     a_scale == nullptr || (glnn::aligned16(a_scale) && glnn::aligned16(a_shift)), relu != 0);
  return glnn::check_launch("glnn_act_fwd_f32");
}

extern "C" int glnn_act_fwd_f32(const float* z, int64_t ldz, int64_t rows, int h, const float* a_scale, const float* a_shift,
                                float drop_p, uint32_t drop_seed, float* y, int64_t ldy, void* stream) {
  return act_fwd_impl(z, ldz, rows, h, a_scale, a_shift, 1, drop_p, drop_seed, y, ldy, stream);
}

// y = dropout(relu?(z * a_scale + a_shift)): glnn_act_fwd_f32 with the ReLU optional (relu = 0: the norm -> dropout tail of
// GCN.forward, reference models.py:189-199)
extern "C" int glnn_norm_drop_fwd_f32(const float* z, int64_t ldz, int64_t rows, int h, const float* a_scale, const float* a_shift,
                                      int relu, float drop_p, uint32_t drop_seed, float* y, int64_t ldy, void* stream) {
  return act_fwd_impl(z, ldz, rows, h, a_scale, a_shift, relu, drop_p, drop_seed, y, ldy, stream);
}

// column sums of a [rows, h] matrix: per-128-row-chunk partials (4 row lanes x 64 columns per workgroup), then the fixed-order
// chunk sum the BatchNorm backward uses -- the bias gradient of a layer whose dz is not produced by glnn_bn_relu_bwd_f32
// (the last GraphConv of the full-graph GCN step, reference train_and_eval.py:12-29).
__global__ __launch_bounds__(256) void col_sum_partial_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int h,
                                                              float* __restrict__ ws) {
  const int lc = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lc;
  const int colc = col < h ? col : h - 1;
  const int64_t r0 = (int64_t)blockIdx.y * kBnRows;
  int64_t r1 = r0 + kBnRows;
  if (r1 > rows) r1 = rows;
  float s = 0.f;
  {                                  // the lane's 32 rows in four batches of eight loads (same order of addition; the plain loop was one
    float t[8];                      // memory round trip per row)
    for (int64_t rb = r0 + rl; rb < r1; rb += 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = x[(rb + 4 * u < r1 ? rb + 4 * u : r0) * ldx + colc];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += rb + 4 * u < r1 ? t[u] : 0.f;
    }
  }
  __shared__ float sh[4][64];
  sh[rl][lc] = s;
  __syncthreads();
  if (rl == 0 && col < h) ws[(int64_t)blockIdx.y * h + col] = (sh[0][lc] + sh[1][lc]) + (sh[2][lc] + sh[3][lc]);
}

extern "C" int glnn_col_sum_f32(const float* x, int64_t ldx, int64_t rows, int h, float* out, float* workspace,
                                int64_t workspace_floats, void* stream) {
  GLNN_REQUIRE(x && out && rows >= 1 && h >= 1 && ldx >= h, "glnn_col_sum_f32: bad arguments");
  const int nchunks = (int)((rows + kBnRows - 1) / kBnRows);
  GLNN_REQUIRE(workspace && workspace_floats >= (int64_t)nchunks * h, "glnn_col_sum_f32: workspace needs >= %lld floats",
               (long long)nchunks * h);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(col_sum_partial_kernel, dim3((h + 63) / 64, nchunks), dim3(256), 0, st, x, ldx, rows, h, workspace);
  if (nchunks > kManyChunks)
    fold_chunks(workspace, nullptr, nchunks, h, out, st);
  else
    hipLaunchKernelGGL(chunk_sum_kernel, dim3((h + 127) / 128), dim3(128), 0, st, workspace, nchunks, h, out);
  return glnn::check_launch("glnn_col_sum_f32");
}

extern "C" int glnn_dropout_mask_u8(int64_t rows, int h, float drop_p, uint32_t drop_seed, uint8_t* mask, void* stream) {
  GLNN_REQUIRE(mask && rows >= 1 && h >= 1 && drop_p >= 0.f && drop_p < 1.f, "glnn_dropout_mask_u8: bad arguments");
  int64_t blocks = (rows * h + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), rows, h,
                     glnn::drop_threshold(drop_p), drop_seed, mask);
  return glnn::check_launch("glnn_dropout_mask_u8");
}

int glnn::adam_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                    const int64_t* sizes, int num_tensors, int64_t max_size, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int64_t step, const float* const* grads_host, const PendingFolds* pending, void* stream) {
  GLNN_REQUIRE(params && grads && exp_avg && exp_avg_sq && sizes, "glnn_adam_step_f32: null pointer");
  GLNN_REQUIRE(num_tensors >= 1 && max_size >= 1 && step >= 1, "glnn_adam_step_f32: bad sizes/step");
  AdamArgs a = {};
  a.p = params; a.g = grads; a.m = exp_avg; a.v = exp_avg_sq; a.sizes = sizes;
  a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay;
  const double bc1 = 1.0 - std::pow((double)beta1, (double)step);
  const double bc2 = 1.0 - std::pow((double)beta2, (double)step);
  a.step_size = (float)((double)lr / bc1);
  a.bc2_sqrt = (float)std::sqrt(bc2);
  a.num_tensors = num_tensors;
  int extra = 0;
  if (pending) {
    GLNN_REQUIRE(grads_host && num_tensors <= kAdamSrcMax, "glnn::adam_step: pending folds need the host pointer table and <= %d tensors", kAdamSrcMax);
    a.nsrc = num_tensors;
    for (int i = 0; i < pending->n; ++i) {
      const GradFold& f = pending->e[i];
      if (f.nslab == 0) continue;
      int t = -1;
      for (int k = 0; k < num_tensors; ++k)
        if (grads_host[k] == f.grad) t = k;
      GLNN_REQUIRE(t >= 0, "glnn::adam_step: a pending gradient fold matches no tensor of the table");
      a.src[t] = {f.src, f.nslab, f.lanes4, f.stride};
    }
    if (pending->has_loss) { a.lf = pending->loss; extra = 1; }
  }
  int64_t chunks = (max_size + (pending ? 255 : 1023)) / (pending ? 256 : 1024);       // folding: one element per thread while the grid allows
  if (chunks > 1024) chunks = 1024;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)chunks, (unsigned)(num_tensors + extra)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), a);
  return glnn::check_launch("glnn_adam_step_f32");
}

extern "C" int glnn_adam_step_f32(float* const* params, const float* const* grads, float* const* exp_avg,
                                  float* const* exp_avg_sq, const int64_t* sizes, int num_tensors, int64_t max_size,
                                  float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                                  void* stream) {
  return glnn::adam_step(params, grads, exp_avg, exp_avg_sq, sizes, num_tensors, max_size, lr, beta1, beta2, eps, weight_decay, step,
                         nullptr, nullptr, stream);
}
