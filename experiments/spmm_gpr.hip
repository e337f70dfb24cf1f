// EXPERIMENT (rounds 3-4; removed from csrc/spmm.hip in round 4 -- it lost its A/B): the "group per row" (GPR) form of the aggregation's
// row role.  Measured equal to the shipping wave-per-row kernel within 1-3 % on every shape (profiles/r03_ab_spmm_gpr.txt): the
// aggregation is bound by the memory system's random-line throughput, not by the per-row latency chain this form removes
// (NOTES.md, "where the aggregation's remaining 30 % is").  Kept here as text for the record; it compiled against csrc/spmm.hip's
// SpmmArgs / wave_gather_sum / long_rows_role / finish helpers at commit 8029310 (round 4) and was selected by GLNN_SPMM_GPR=1.
#if 0
// ---------------------------------------------------------------------------------------------
// K1 "group per row" (GPR) form of the row role, for rows covered by 16 / 32 / 64 lanes (48 < d <= 256) without col_scale.
// Why: on uniform random graphs the wave-per-row kernel above costs  t = a * rows + b * edges  with b = 6.0-6.2 TB/s of
// gathered bytes whatever the width (the part's random-gather ceiling; scripts/probe_gather.py) and a = 0.25-0.65 ns per
// ROW chip-wide = about one fully exposed memory latency per row and wave: the chain indptr -> indices -> gather is paid
// per row, a degree-20 row fills 1 1/4 iterations of G x U lanes, and the memory pipe drains at every row boundary
// (16 % of the XL shard's time, 14 % of the D=47 layer).  Here
//   * a wave takes a BATCH of BR consecutive rows per ticket: their BR+1 indptr entries arrive in ONE coalesced load and
//     stay in a register (lane i = row i), their in-edges are one contiguous slice of `indices`;
//   * every LANE GROUP (LPR lanes) owns one row at a time and walks its edges U per iteration in edge order (no
//     cross-group fold; the G rows of a wave progress concurrently, so a short row no longer leaves 3/4 of the lanes idle);
//   * a group that finishes a row stores it and continues with its NEXT row, which it was dealt when the current one began:
//     that row's first LPR column indices are already in a register (prefetched a row ahead, as is the next index chunk of
//     a row that is longer than LPR), and its self row is requested when the row begins -- no load in the loop is
//     consumed in the iteration that issues it, except the gathers themselves;
//   * rows are dealt dynamically inside the wave (ballot + popcount), so groups stay busy until the batch runs out.
// Per-row summation order: edges in storage order, one accumulator (what the CPU oracle does); rows of degree > LONG_ROW
// are left to the long-row role exactly as before.
// ---------------------------------------------------------------------------------------------
#ifndef GLNN_GPR_BATCH
#define GLNN_GPR_BATCH 32
#endif
#ifndef GLNN_GPR_U
#define GLNN_GPR_U 8
#endif
constexpr int kBatchRows = GLNN_GPR_BATCH;      // <= 63: lane i of the wave holds indptr[v0 + i]
static_assert(kBatchRows >= 4 && kBatchRows <= 63, "batch rows");

template <int MODE>
__device__ __forceinline__ void finish_row_gpr(const SpmmArgs& a, int64_t v, int deg, float4 acc, float4 selfv, int col4, const EpCols& ep) {
  float4 y;
  if (MODE == GLNN_AGG_SAGE_GCN) {
    y = mean4(acc, selfv, (float)deg + 1.0f);
  } else {
    const float rs = a.row_scale ? a.row_scale[v] : 1.0f;
    y = make_float4(acc.x * rs, acc.y * rs, acc.z * rs, acc.w * rs);
  }
  float yy[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int c = col4 + t;
    if (c < a.d) {
      if (a.ep_scale) yy[t] *= ep.s[t];
      if (a.ep_shift) yy[t] += ep.h[t];
      if (a.relu) yy[t] = fmaxf(yy[t], 0.f);
    } else {
      yy[t] = 0.f;
    }
  }
  st4_stream(a.out + v * a.ldo + col4, make_float4(yy[0], yy[1], yy[2], yy[3]));
}

template <int LPR, int U, int MODE>
__global__ __launch_bounds__(kBlock) void spmm_gpr_kernel(const SpmmArgs a) {
  constexpr int G = 64 / LPR;
  static_assert(LPR % U == 0, "an index chunk (LPR entries) is consumed in whole iterations");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane % LPR, gl0 = lane - li;
  const int col4 = li * 4;
  const bool col_ok = col4 < a.d;
  const EpCols ep = load_ep_cols(a, col_ok ? col4 : 0);

  if ((int)blockIdx.x < a.n_long_blocks) {
    long_rows_role<LPR, U, MODE, false>(a, lane, wave, col4, col_ok, ep);
    return;
  }
  __shared__ int s_ticket;
  if (threadIdx.x == 0) s_ticket = 0;
  __syncthreads();
  const int64_t blk = (int64_t)blockIdx.x - a.n_long_blocks;
  const int64_t row_base = blk * a.rows_per_block;
  const int n_batches = a.rows_per_block / kBatchRows;
  const unsigned long long lower = (1ull << gl0) - 1ull;          // lanes of the groups in front of mine
  const bool self_early = (MODE == GLNN_AGG_SAGE_GCN) && a.self_rows == nullptr;
#pragma unroll 1
  while (true) {
    int t = 0;
    if (lane == 0) t = atomicAdd(&s_ticket, 1);
    t = __builtin_amdgcn_readfirstlane(t);
    if (t >= n_batches) break;
    const int64_t v0 = row_base + (int64_t)t * kBatchRows;
    if (v0 >= a.n_dst) break;
    const int64_t left = a.n_dst - v0;
    const int nrows = left < kBatchRows ? (int)left : kBatchRows;
    // the batch's row offsets, relative to its first edge: lane i <- indptr[v0 + i] - indptr[v0]
    const int64_t my_ip = a.indptr[v0 + (lane <= nrows ? lane : nrows)];
    const int64_t e_first = __shfl(my_ip, 0);
    const int ipl = (int)(my_ip - e_first);
    const int32_t* __restrict__ idx = a.indices + e_first;
    const int bend = __shfl(ipl, nrows);                              // edges of the batch
    // ---- per-group state (identical in the LPR lanes of a group) ----
    int r = lane / LPR, rn = G + lane / LPR, next_row = 2 * G;
    bool act = r < nrows;
    int pos = __shfl(ipl, r < nrows ? r : nrows), end = __shfl(ipl, r + 1 < nrows ? r + 1 : nrows);
    int deg = end - pos;
    bool skip = deg > kLongRow;
    if (skip) end = pos;
    int cb = pos;                                                     // base of the current index chunk
    int cidx = (act && cb + li < bend) ? ld_idx_stream(idx + cb + li) : 0;
    int pidx;                                                         // the chunk after it, in this group's stream order
    {
      const bool in_row = cb + LPR < end;
      const int nb = __shfl(ipl, rn < nrows ? rn : nrows);          // (not inside the ?: -- a shuffle must run with every lane active)
      const int sb = in_row ? cb + LPR : nb;
      pidx = (act && sb + li < bend) ? ld_idx_stream(idx + sb + li) : 0;
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), selfv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (self_early && act && col_ok) selfv = ld4(a.x_self + (v0 + r) * a.ld_self + col4);
#pragma unroll 1
    while (__ballot(act) != 0ull) {
      float4 v[U];
      const int rel = gl0 + (pos - cb);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int src;
        if (G == 1) src = __builtin_amdgcn_readlane(cidx, __builtin_amdgcn_readfirstlane(rel) + u);
        else src = __shfl(cidx, rel + u);
        const bool ok = act && (pos + u < end) && col_ok;
        v[u] = ok ? ld4(a.x + (int64_t)src * a.ldx + col4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc = add4(acc, v[u]);
      pos += U;
      const bool row_done = act && pos >= end;
      const bool chunk_done = act && !row_done && (pos - cb) >= LPR;
      if (__ballot(row_done) != 0ull) {
        if (row_done && !skip && col_ok) {
          const int64_t vv = v0 + r;
          if (MODE == GLNN_AGG_SAGE_GCN && !self_early) selfv = ld4(a.x_self + a.self_rows[vv] * a.ld_self + col4);
          finish_row_gpr<MODE>(a, vv, deg, acc, selfv, col4, ep);
        }
        // deal the next rows: the finishing groups, in lane order, get next_row, next_row + 1, ...  (the shuffles run with
        // ALL lanes active -- this branch is wave-uniform -- because ds_bpermute reads nothing from a masked-off lane)
        const unsigned long long fin = __ballot(row_done && li == 0);
        const int rank = __popcll(fin & lower);
        const int r_new = rn;
        const int p_new = __shfl(ipl, r_new < nrows ? r_new : nrows), e_new = __shfl(ipl, r_new + 1 < nrows ? r_new + 1 : nrows);
        if (row_done) {
          r = r_new;
          rn = next_row + rank;
          act = r < nrows;
          pos = p_new;
          end = e_new;
          deg = end - pos;
          skip = deg > kLongRow;
          if (skip) end = pos;
          cb = pos;
          cidx = pidx;
          acc = make_float4(0.f, 0.f, 0.f, 0.f);
          if (self_early && act && col_ok) selfv = ld4(a.x_self + (v0 + r) * a.ld_self + col4);
        }
        next_row += __popcll(fin);
      }
      if (chunk_done) { cb += LPR; cidx = pidx; }
      const bool adv = row_done || chunk_done;
      if (__ballot(adv) != 0ull) {
        const bool in_row = cb + LPR < end;
        const int nb = __shfl(ipl, rn < nrows ? rn : nrows);
        const int sb = in_row ? cb + LPR : nb;
        if (adv) pidx = (act && sb + li < bend) ? ld_idx_stream(idx + sb + li) : 0;
      }
    }
  }
}

template <int LPR, int U>
int launch_gpr(const SpmmArgs& a, int mode, hipStream_t st, int grid) {
  if (mode == GLNN_AGG_SAGE_GCN) {
    hipLaunchKernelGGL((spmm_gpr_kernel<LPR, U, GLNN_AGG_SAGE_GCN>), dim3(grid), dim3(kBlock), 0, st, a);
  } else {
    hipLaunchKernelGGL((spmm_gpr_kernel<LPR, U, GLNN_AGG_SUM>), dim3(grid), dim3(kBlock), 0, st, a);
  }
  return glnn::check_launch("glnn_spmm_csr_f32");
}

    // group-per-row form (see spmm_gpr_kernel): rows of 16 / 32 / 64 lanes, no col_scale.  OPT-IN (GLNN_SPMM_GPR=1): measured
    // equal to the wave-per-row kernel within 1-3 % on every shape (profiles/r03_ab_spmm_gpr.txt) -- the aggregation is bound by
    // the memory system's random-line throughput, not by the per-row latency chain this form removes
    const char* ge = getenv("GLNN_SPMM_GPR");                 // read per call (a ~100 ns lookup): A/B scripts flip it inside one process
    const int gpr_env = ge ? atoi(ge) : 0;
    const char* gm = getenv("GLNN_SPMM_GPR_MIN_ROWS");
    const int64_t gpr_min_rows = gm ? atoll(gm) : 131072;
    if (gpr_env && !wide && !col_scale && dv > 8 && n_dst >= gpr_min_rows) {
      int64_t nb = n_dst / (2048 * kBatchRows);            // batches per workgroup: two per wave on whole graphs
      if (nb < 1) nb = 1;
      if (nb > 2 * kWavesPerBlock) nb = 2 * kWavesPerBlock;
      a.rows_per_block = (int)(nb * kBatchRows);
      const int64_t rb = (n_dst + a.rows_per_block - 1) / a.rows_per_block;
      GLNN_REQUIRE(rb + n_long < ((int64_t)1 << 31), "glnn_spmm_csr_f32: n_dst too large for one launch");
      const int g2 = (int)(rb + n_long);
      if (dv <= 16) rc = launch_gpr<16, GLNN_GPR_U>(a, mode, st, g2);
      else if (dv <= 32) rc = launch_gpr<32, GLNN_GPR_U>(a, mode, st, g2);
      else rc = launch_gpr<64, GLNN_GPR_U>(a, mode, st, g2);
      if (rc != GLNN_OK) return rc;
      continue;
    }
#endif
