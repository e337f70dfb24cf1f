// Device-side pieces shared by student.hip and mlp_lat.hip: the "last workgroup finishes the reduction" protocol, the
// BatchNorm statistics combine and the argument blocks of the loss / statistics kernels.  Included inside each translation
// unit's anonymous namespace users; everything here is static to the including file.
#pragma once
#include <cmath>

#include "glnn_common.h"

namespace {


__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// "Last workgroup finishes the reduction": every workgroup publishes its partials, bumps a counter, and the one that sees
// the final count folds all partials in fixed order -- the second (finalize) launch of a two-stage reduction disappears
// (each launch costs ~5 us of device-side latency in the small-batch student step, whatever it computes).  The counter
// must be 0 on entry and is reset by the last workgroup; callers that cannot guarantee that (the plain C entry points,
// whose workspaces are uninitialised) pass NULL and get the two-launch form.
// Cross-workgroup visibility WITHOUT a device-scope fence (which writes back / invalidates the XCD's whole L2 -- measured:
// the fenced form made the MLP3w8 step 25 % SLOWER): the partials are published with relaxed agent-scope atomic stores
// (write-through past the per-XCD L2) and read back by the last workgroup with relaxed agent-scope atomic loads.
// ORDER: a relaxed store followed by __syncthreads() is NOT enough -- the workgroup-scope fence of the barrier does not
// wait for the write-through (the gfx950 ISA had `global_store_dword ... sc1; s_barrier; global_atomic_add` with no
// s_waitcnt vmcnt(0) in between), so another XCD could see the final count before the partials.  Every storing wave
// therefore drains its own vector-memory queue (publish_drain) before the barrier in front of the counter update;
// tests/test_capi_symbols.py asserts the s_waitcnt vmcnt(0) on the generated ISA.
__device__ __forceinline__ void st_part(float* p, float v, bool shared) {
  if (shared) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
__device__ __forceinline__ float ld_part(const float* p) {
  return __hip_atomic_load(const_cast<float*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// all of this wave's global stores (incl. the write-through partials) have been acknowledged by the memory system
__device__ __forceinline__ void publish_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ bool last_workgroup(int* counter, int total) {
  __shared__ int s_last;
  publish_drain();                                   // this wave's partial stores have been acknowledged ...
  __syncthreads();                                   // ... and so have every other wave's of the workgroup
  if (threadIdx.x == 0) {
    const int prev = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (prev == total - 1);
    if (s_last) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  return s_last != 0;
}


struct LossArgs {
  const float* z; int64_t ldz; int64_t rows; int c; int kind;
  const int64_t* labels; const int64_t* label_rows;
  const float* t; int64_t ldt; const int64_t* t_rows;
  float scale;  // lamb / rows
  float* dz; int64_t ldg; float* logp; int64_t ldl;
  float* partial;  // [gridDim.x] per-block loss sums, or NULL (log_softmax only)
  // fused finalize (counter != NULL): the last workgroup writes the loss (and, with col_sum != NULL and c <= 64, the
  // column sums of dz = the bias gradient of the layer that produced the logits; col_partial [gridDim.x][64] scratch)
  int* counter; float inv_rows; float* loss_out; float* loss_accum; float* col_sum; float* col_partial;
  // nslab > 0 (c <= 64): the logits are still split-K partials -- logit(row, j) = sum_s slabs[s * slab_stride + row * c + j] + bias[j],
  // summed here (s ascending) and stored to z (the fold launch of the producing GEMM is gone)
  const float* slabs; int nslab; int64_t slab_stride; const float* bias; float* z_store;
  // defer != 0: the partials (loss per workgroup; with col_sum != NULL the [blocks][64] column partials) are left in memory for the
  // fused Adam launch to fold (glnn::PendingFolds): plain stores, no counter, no last-workgroup tail
  int defer;
};

// last workgroup of a loss launch: loss = sum of the per-workgroup partials / rows (fixed order) and, with col_sum, the column
// sums of dz from the [nblocks][64] column partials (4 partial lanes per column, folded in fixed order).  All 256 threads call it.
__device__ __forceinline__ void loss_fold_last(const LossArgs& a, int nblocks, float (&sc)[4][64]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ float red[256];
  float v = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += 256) v += ld_part(&a.partial[i]);
  red[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float l = red[0] * a.inv_rows;
    if (a.loss_out) a.loss_out[0] = l;
    if (a.loss_accum) a.loss_accum[0] += l;
  }
  if (a.col_sum) {
    float cs = 0.f;
#pragma unroll 8
    for (int k = wave; k < nblocks; k += 4) cs += ld_part(&a.col_partial[(int64_t)k * 64 + lane]);
    __syncthreads();
    sc[wave][lane] = cs;
    __syncthreads();
    if ((int)threadIdx.x < a.c) a.col_sum[threadIdx.x] = (sc[0][lane] + sc[1][lane]) + (sc[2][lane] + sc[3][lane]);
  }
}

constexpr int kBnRows = 128;  // rows per chunk

// Column kernels below share one mapping: a workgroup owns 64 columns x one chunk of kBnRows rows; thread =
// (column = tid & 63, row lane = tid >> 6); a row lane walks rows r0 + lane + 4*i.  Loads are issued 8 rows at a
// time from clamped (always valid) addresses so that they pipeline instead of serialising on a loop-carried
// dependence; rows past the chunk end are masked arithmetically.
constexpr int kRowLanes = 4;
constexpr int kRowsPerLane = kBnRows / kRowLanes;   // 32
constexpr int kUnroll = 8;
constexpr int kBnGroup = 64;      // partial triples per first-level group of the two-level statistics combine
constexpr int kManyChunks = 64;   // row chunks above which per-chunk partials are folded by a lane-split pass (> 8192 rows)

struct BnFinArgs {
  // partial k of column c:  mean = ws_mean[k*pstride + c], M2 = ws_m2[k*pstride + c], count = ws_cnt ? ws_cnt[k*pstride + c]
  // : rows of chunk k.  (Local chunks: pstride = h, ws_cnt = NULL.  Gathered per-rank triples: pstride = 3h.)
  const float* ws_cnt; const float* ws_mean; const float* ws_m2; int nparts; int64_t pstride; int64_t rows; int h;
  // emit mode (emit_cnt != NULL): write the combined (count, mean, M2) and stop -- the per-rank triple that is exchanged
  float* emit_cnt; float* emit_mean; float* emit_m2;
  // group_len > 0 (emit mode only): workgroup (x, y) combines the partials [y*group_len, (y+1)*group_len) and emits triple y
  // (row y of the emit arrays): the first level of a two-level combine for thousands of row chunks
  int group_len;
  const float* gamma; const float* beta; float eps; float momentum;
  float* running_mean; float* running_var; int64_t* nbt;
  float* mean_out; float* rstd_out; float* a_scale; float* a_shift; float* rows_out;
  int chunk_rows;      // rows per local chunk partial (0 = kBnRows; 32 = the tiles of mlp_lat.hip's GEMM epilogue)
};

__device__ __forceinline__ double part_count(const BnFinArgs& a, int k, int col) {
  if (a.ws_cnt) return (double)a.ws_cnt[(int64_t)k * a.pstride + col];
  const int cr = a.chunk_rows > 0 ? a.chunk_rows : kBnRows;
  int64_t r0 = (int64_t)k * cr, r1 = r0 + cr;
  if (r1 > a.rows) r1 = a.rows;
  return (double)(r1 - r0);
}

// The end of the statistics of one column from its combined (N, mean, M2): normalisation constants, the fused forward transform
// y = z * a_scale + a_shift, the running statistics (momentum, unbiased variance) -- stored only when `store`; sc / sh returned.
// g / b: gamma[col] / beta[col] (1 / 0 without affine), loaded by the caller BEFORE it waits for the partials -- loaded here they are
// one more dependent round trip to memory at the end of a latency chain.
__device__ __forceinline__ void bn_emit_column(const BnFinArgs& a, int col, double n, double mean, double m2, bool store, float g, float b,
                                               float& sc, float& sh) {
  const float var_b = (float)(m2 / n);                         // biased: used for normalisation
  const float var_u = n > 1.0 ? (float)(m2 / (n - 1.0)) : var_b;  // unbiased: running_var
  const float meanf = (float)mean;
  const float rstd = 1.0f / sqrtf(var_b + a.eps);
  sc = g * rstd;
  sh = b - meanf * sc;
  if (!store) return;
  if (col == 0 && a.rows_out) a.rows_out[0] = (float)n;
  if (a.mean_out) a.mean_out[col] = meanf;
  if (a.rstd_out) a.rstd_out[col] = rstd;
  a.a_scale[col] = sc;
  a.a_shift[col] = sh;
  if (a.running_mean) a.running_mean[col] = (1.f - a.momentum) * a.running_mean[col] + a.momentum * meanf;
  if (a.running_var) a.running_var[col] = (1.f - a.momentum) * a.running_var[col] + a.momentum * var_u;
}

// One workgroup = 64 columns x 4 partial lanes (the column kernels' mapping): lane rl combines partials rl, rl+4, ... ; the
// four lane results are combined in fixed order through LDS.  (A single thread per column walked the nparts partials as
// one chain of dependent L2 round trips: 13-21 us for 32 partials.)
template <bool COH>      // COH: the partials were published by other workgroups of THIS launch -> read them past the L2
__device__ __forceinline__ void bn_finalize_columns(const BnFinArgs& a, int colblock) {
  const int lc = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int col = colblock * 64 + lc;
  const int colc = col < a.h ? col : a.h - 1;
  if (colblock == 0 && threadIdx.x == 0 && a.nbt && !a.emit_cnt) a.nbt[0] += 1;
  const int p0 = a.group_len > 0 ? (int)blockIdx.y * a.group_len : 0;
  const int p1 = a.group_len > 0 ? (p0 + a.group_len < a.nparts ? p0 + a.group_len : a.nparts) : a.nparts;
  const int64_t eoff = a.group_len > 0 ? (int64_t)blockIdx.y * a.h : 0;
  __shared__ double sh_n[kRowLanes][64], sh_s[kRowLanes][64];
  // combine of the partial (count, mean, M2) triples in double, fixed order, ONE pass over the partials (both arrays of a partial are
  // requested together; the former two-pass form -- means first, then M2 around the combined mean -- was two dependent round trips
  // to memory in the tail of every launch that ends with it):
  //   N = sum n_k ;  S = sum n_k*mean_k ;  Q = sum [ M2_k + n_k*mean_k^2 ] ;  mean = S / N ;  M2 = Q - S*mean
  // In double the subtraction costs ~1e-16 * N*mean^2 against M2: invisible after rounding to float unless |mean|/std > 1e6.
  const float gam = a.gamma ? a.gamma[colc] : 1.f, bet = a.beta ? a.beta[colc] : 0.f;
  double n = 0.0, sum = 0.0, qs = 0.0;
  // eight partials of this row lane per batch: all their loads are issued before the first is used (the plain loop compiled to
  // load, wait, accumulate, branch -- one memory round trip per partial at the end of a latency chain); same order of accumulation
  for (int k0 = p0 + rl; k0 < p1; k0 += 8 * kRowLanes) {
    float mk[8], m2k[8], ck[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + u * kRowLanes < p1 ? k0 + u * kRowLanes : p1 - 1;
      mk[u] = COH ? ld_part(&a.ws_mean[(int64_t)k * a.pstride + colc]) : a.ws_mean[(int64_t)k * a.pstride + colc];
      m2k[u] = COH ? ld_part(&a.ws_m2[(int64_t)k * a.pstride + colc]) : a.ws_m2[(int64_t)k * a.pstride + colc];
      ck[u] = a.ws_cnt ? a.ws_cnt[(int64_t)k * a.pstride + colc] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + u * kRowLanes;
      if (k < p1) {
        const double nb = a.ws_cnt ? (double)ck[u] : part_count(a, k, colc);
        n += nb;
        sum += nb * (double)mk[u];
        qs += (double)m2k[u] + nb * (double)mk[u] * (double)mk[u];       // nb == 0: an empty slice contributes nothing
      }
    }
  }
  __shared__ double sh_q[kRowLanes][64];
  sh_n[rl][lc] = n;
  sh_s[rl][lc] = sum;
  sh_q[rl][lc] = qs;
  __syncthreads();
  if (rl != 0 || col >= a.h) return;
  n = (sh_n[0][lc] + sh_n[1][lc]) + (sh_n[2][lc] + sh_n[3][lc]);
  sum = (sh_s[0][lc] + sh_s[1][lc]) + (sh_s[2][lc] + sh_s[3][lc]);
  qs = (sh_q[0][lc] + sh_q[1][lc]) + (sh_q[2][lc] + sh_q[3][lc]);
  const double mean = sum / n;
  double m2 = qs - sum * mean;
  if (m2 < 0.0) m2 = 0.0;
  if (a.emit_cnt) {
    a.emit_cnt[eoff + col] = (float)n;
    a.emit_mean[eoff + col] = (float)mean;
    a.emit_m2[eoff + col] = (float)m2;
    return;
  }
  float sc, sh;
  bn_emit_column(a, col, n, mean, m2, true, gam, bet, sc, sh);
}

}  // namespace
