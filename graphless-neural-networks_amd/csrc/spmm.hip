// K1/K2: CSR neighbour aggregation for gfx950 (MI355X).  HBM-bound gather: the only goal is to keep
// as many 16-byte-per-lane row loads in flight as the memory system will take.
//
// Replaces (see include/glnn_hip.h): dgl SAGEConv "gcn" aggregation (reference models.py:112,138),
// dgl GraphConv norm="both" aggregation (models.py:193) and update_all(copy_u,sum) (utils.py:185).
//
// Mapping
//   * one 64-lane wavefront per destination row (rows of degree <= LONG_ROW), 8 waves per workgroup pulling the
//     workgroup's rows from an LDS ticket counter (dynamic balance inside the workgroup);
//   * a feature row of d floats is covered by LPR = 4..64 lanes moving float4 (LPR*16 B per row);
//     the G = 64/LPR lane groups of the wave take different in-edges of the same row, U edges per
//     group in flight, and are folded with two cross-lane adds at the end;
//   * the row's column indices are read coalesced (one per lane, 64 at a time) and handed to the
//     groups with ds_bpermute / v_readlane -- no per-edge dependent index load;
//   * power-law tails: rows with degree > LONG_ROW are skipped by the row waves and taken by the
//     first `n_long_blocks` workgroups of the SAME launch (dispatched first, so their long latency
//     overlaps the bulk), 8 waves per row, LDS fold in fixed order => deterministic results;
//   * epilogue fused: +self, /(deg+1) (SAGE "gcn") or *row_scale (GraphConv / feature_prop), then
//     optional per-column scale/shift (+ReLU) for the project-first form.
#include "glnn_common.h"

namespace {

// tuning knobs (compile-time).  Defaults = best of an interleaved sweep on ogbn-products shape, one MI355X:
//   long-row threshold 64/96/128/192/256/512/2048 -> 128 (fused 100->256: 10.75 -> 9.8 ms, 256->256: 23.5 -> 22.3 ms
//   vs 512; 64 over-uses the cooperative path); rows per wave 4/8/16 -> 16; stand-alone U 4/8 -> 8 (D=100: 9.45 ->
//   8.8 ms); fused U 4/6/8/12 -> 8 (4..8 equal, 12 loses occupancy)
#ifndef GLNN_SPMM_U
#define GLNN_SPMM_U 8
#endif
#ifndef GLNN_FUSED_U
#define GLNN_FUSED_U 8
#endif
#ifndef GLNN_ROWS_PER_WAVE
#define GLNN_ROWS_PER_WAVE 16
#endif
#ifndef GLNN_LONG_ROW
#define GLNN_LONG_ROW 128
#endif
#ifndef GLNN_LONG_BLOCK_ROWS
#define GLNN_LONG_BLOCK_ROWS 512
#endif
#ifndef GLNN_LONG_BLOCK_CAP
#define GLNN_LONG_BLOCK_CAP 512
#endif
constexpr int kBlock = 512;              // 8 waves
constexpr int kWavesPerBlock = kBlock / 64;
constexpr int kRowsPerWave = GLNN_ROWS_PER_WAVE;
constexpr int kLongRow = GLNN_LONG_ROW;  // degree above which a whole workgroup takes the row
// HUB rows (round 5): a row of more than kHubRow in-edges is summed in SEGMENTS of kHubSeg = 512 edges, 64 per wave: wave w's share of the row
// is the sum over the segments s (ascending) of P(s, w), its 64-edge piece of segment s gathered into a fresh accumulator; the eight shares
// are folded like the eight wave partials of any long row.  So the pieces can be gathered by OTHER workgroups (hub_gather_kernel, one
// workgroup per segment, into a slab the row's owner reads back) without changing a bit: with or without a plan, whole graph, chunk or
// shard, a hub row's sum is the same number -- and without a plan the owner pays no barrier for it.
// Why: a shard's chunk launch takes ~0.7 ms, one workgroup needs ~0.5 ms for a 17 k-edge row at D = 256 (scripts/hub_tail_probe.py).
#ifndef GLNN_HUB_ROW
#define GLNN_HUB_ROW 1024
#endif
constexpr int kHubRow = GLNN_HUB_ROW;
constexpr int kHubSeg = 512;

// the hub rows of one launch and where their segments' partial sums are (glnn_hub_plan of the C ABI, checked by the launcher)
struct HubPlan {
  const int64_t* rows;          // ascending row ids (relative to the launch's indptr), degree > kHubRow
  const int32_t* seg_ptr;       // [n_hub + 1] first segment of each hub row; seg_ptr[n_hub] = number of segments
  int n_hub;
  float* slab;                  // [8 n_seg][ld_slab] the wave partials P(s, w) at row 8 s + w, written by hub_gather_kernel
  int64_t ld_slab;
};

// Round 6: ONE launch over the CHUNKS of a rank's rows (glnn_sage_fused_chunks_f32).  The rows of chunk c -- tiles [tile_start[c],
// tile_start[c + 1]) -- read their self rows and write their outputs at a per-chunk row shift (the chunk-major buffers of the sharded
// forward keep a rank's chunks in separate slots), and every workgroup ARRIVES on its chunk's counter when its waves' stores are done: stores ->
// (acknowledged by the L2) -> atomic add; the wave that completes the count stores the launch's epoch to the chunk's
// signal word, on which the host's exchange stream waits (hipStreamWaitValue32) before it sends the chunk -- while the same launch is
// still working on the next chunk.
struct ChunkMap {
  int n;
  int tile_start[GLNN_MAX_CHUNKS + 1];
  int64_t self_shift[GLNN_MAX_CHUNKS];                  // self row of own row v = x_self + (v + self_shift[c]) * ld_self
  int64_t out_shift[GLNN_MAX_CHUNKS];                   // output row of own row v = v + out_shift[c] (out and out2)
  int* arrivals;                                        // [n] counters, zero between launches (the completing wave resets its counter)
  uint32_t* signal[GLNN_MAX_CHUNKS];
  uint32_t epoch;
};

// constant-index selects over the by-value arrays of a ChunkMap (a dynamic index would send the kernel arguments through scratch)
__device__ __forceinline__ int64_t cm_pick(const int64_t (&a)[GLNN_MAX_CHUNKS], int c) {
  int64_t r = a[0];
#pragma unroll
  for (int i = 1; i < GLNN_MAX_CHUNKS; ++i) r = (c == i) ? a[i] : r;
  return r;
}
__device__ __forceinline__ int cm_tile_start(const ChunkMap& cm, int c) {       // c in [0, GLNN_MAX_CHUNKS]
  int r = cm.tile_start[0];
#pragma unroll
  for (int i = 1; i <= GLNN_MAX_CHUNKS; ++i) r = (c == i) ? cm.tile_start[i] : r;
  return r;
}
__device__ __forceinline__ uint32_t* cm_signal(const ChunkMap& cm, int c) {
  uint32_t* r = cm.signal[0];
#pragma unroll
  for (int i = 1; i < GLNN_MAX_CHUNKS; ++i) r = (c == i) ? cm.signal[i] : r;
  return r;
}
// one arrival on chunk c's counter (a single thread); the arrival that completes `expected` resets the counter and stores the epoch
__device__ __forceinline__ void cm_arrive(const ChunkMap& cm, int c, int expected) {
  int* cnt = cm.arrivals + c;
  const int prev = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (prev == expected - 1) {
    __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(cm_signal(cm, c), cm.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

struct SpmmArgs {
  const int64_t* indptr;
  const int32_t* indices;
  int64_t n_dst;
  const float* x;
  int64_t ldx;
  int d;
  const float* row_scale;
  const float* col_scale;
  const float* x_self;
  int64_t ld_self;
  const int64_t* self_rows;     // optional: the self row of destination v is x_self[self_rows[v]] (global feature matrix)
  const float* ep_scale;
  const float* ep_shift;
  int relu;
  float* out;
  int64_t ldo;
  int n_long_blocks;
  int rows_per_block;           // rows a row-role workgroup pulls from its ticket (8 waves x 1..kRowsPerWave rows)
  // XF kernels only: every gathered row and every self row is  drop(relu(z * xf_scale + xf_shift))  of the stored row z -- the tail of the
  // hidden layer in front (BatchNorm affine, ReLU, counter-based dropout keyed by the SOURCE row id), evaluated in the gather instead of
  // being written by a pass of its own (glnn_act_fwd_f32's arithmetic, element for element)
  int xf_on; const float* xf_scale; const float* xf_shift; uint32_t xf_thr; uint32_t xf_seed; float xf_dscale;
  HubPlan hub;                  // n_hub == 0: hub rows are summed by the row's own workgroup (same order, same bits)
  ChunkMap cm;                  // n == 0: off (glnn_spmm_csr_chunks_f32: the chunks of a row range in ONE launch, a completion signal per chunk)
};

// per-lane constants of the source transform: the lane's four columns
struct XfCols { float s[4], h[4]; uint32_t thr, seed; float dscale; int col; bool affine; };
__device__ __forceinline__ XfCols load_xf_cols(const SpmmArgs& a, int col4) {
  XfCols x;
  x.thr = a.xf_thr; x.seed = a.xf_seed; x.dscale = a.xf_dscale; x.col = col4; x.affine = a.xf_scale != nullptr;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const bool ok = x.affine && col4 + t < a.d;
    x.s[t] = ok ? a.xf_scale[col4 + t] : 1.f;
    x.h[t] = ok ? a.xf_shift[col4 + t] : 0.f;
  }
  return x;
}
__device__ __forceinline__ float4 xf_apply(const XfCols& x, float4 v, uint32_t row) {
  float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    float y = x.affine ? fmaf(o[t], x.s[t], x.h[t]) : o[t];
    y = fmaxf(y, 0.f);
    if (x.thr) y = glnn::drop_keep(x.seed, x.thr, row, (uint32_t)(x.col + t)) ? y * x.dscale : 0.f;
    asm volatile("" : "+v"(y));        // the ROUNDED tail value is what gets summed (act_fwd stored it): no fma of y * dscale into the row sum
    o[t] = y;
  }
  return make_float4(o[0], o[1], o[2], o[3]);
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// stream-once data (column indices, the output rows) moved with the non-temporal hint so that it does not evict the
// re-used feature rows from L2 / Infinity Cache (GLNN_NT_STREAMS=0 compiles the plain forms for A/B)
#ifndef GLNN_NT_STREAMS
#define GLNN_NT_STREAMS 1
#endif
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4_stream(float* p, float4 v) {
#if GLNN_NT_STREAMS
  f32x4_t t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<f32x4_t*>(p));
#else
  st4(p, v);
#endif
}
__device__ __forceinline__ int ld_idx_stream(const int32_t* p) {
#if GLNN_NT_STREAMS
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
// (a + s) / d for the SAGE-"gcn" mean, d = deg + 1 (an integer below 2^24, exactly representable): reciprocal, quotient, ONE
// residual correction (13 instructions for the four components).  The compiler's IEEE fdiv expansion is ~10 dependent
// instructions PER component (v_div_scale x2, v_rcp, four FMAs, v_div_fmas through VCC, v_div_fixup); four of them per output
// row cost the fused 256-wide layer 7 % (19.9 -> 18.3 ms, interleaved A/B with the plain reciprocal form).
// The corrected quotient is the correctly rounded one except for rare last-bit ties; every aggregation site uses this same
// function, so the kernels stay bit-identical to each other (fused == stand-alone, chunked == whole graph).
__device__ __forceinline__ float div_corrected(float a, float d, float rd) {
  const float q = a * rd;
  const float e = fmaf(-q, d, a);
  return fmaf(e, rd, q);
}
__device__ __forceinline__ float4 mean4(float4 acc, float4 s, float d) {
  const float rd = __builtin_amdgcn_rcpf(d);
  return make_float4(div_corrected(acc.x + s.x, d, rd), div_corrected(acc.y + s.y, d, rd), div_corrected(acc.z + s.z, d, rd),
                     div_corrected(acc.w + s.w, d, rd));
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 fma4(float s, float4 v, float4 a) {
  return make_float4(fmaf(s, v.x, a.x), fmaf(s, v.y, a.y), fmaf(s, v.z, a.z), fmaf(s, v.w, a.w));
}
__device__ __forceinline__ float4 shfl_xor4(float4 v, int m) {
  return make_float4(__shfl_xor(v.x, m), __shfl_xor(v.y, m), __shfl_xor(v.z, m), __shfl_xor(v.w, m));
}

// Sum of x[indices[e], col4..col4+3] over the edges e in [e0,e1) that belong to this wave:
// 64-edge chunks  e0 + 64*(wave_id + k*n_waves).  Returns the total in every lane of group 0
// (lanes < LPR); other lanes hold partial garbage.
// the two halves of wave_gather_sum: the per-GROUP running sums (continuing `acc`: group g of the lanes takes the edges e with
// (e - e0) % G == g in ascending order), and the fold of the G group sums
template <int LPR>
__device__ __forceinline__ float4 fold_groups(float4 acc) {
  constexpr int G = 64 / LPR;
  if (G >= 2) acc = add4(acc, shfl_xor4(acc, 32));
  if (G >= 4) acc = add4(acc, shfl_xor4(acc, 16));
  if (G >= 8) acc = add4(acc, shfl_xor4(acc, 8));
  if (G >= 16) acc = add4(acc, shfl_xor4(acc, 4));
  return acc;
}
template <int LPR, int U, bool CS, bool XF = false>
__device__ __forceinline__ float4 wave_gather_acc(const int32_t* __restrict__ indices, int64_t e0, int64_t e1,
                                                  int wave_id, int n_waves, const float* __restrict__ x,
                                                  int64_t ldx, int col4, bool col_ok,
                                                  const float* __restrict__ col_scale, int lane, const XfCols& xf, float4 acc) {
  constexpr int G = 64 / LPR;
  const int g = lane / LPR;
  for (int64_t base = e0 + (int64_t)wave_id * 64; base < e1; base += (int64_t)n_waves * 64) {
    const int64_t rem = e1 - base;
    const int cnt = rem < 64 ? (int)rem : 64;
    const int my_idx = lane < cnt ? ld_idx_stream(indices + base + lane) : 0;
    float my_cs = 1.f;
    if (CS) my_cs = lane < cnt ? col_scale[my_idx] : 0.f;
    for (int j = 0; j < cnt; j += G * U) {
      float4 v[U];
      float s[U];
      int srcs[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ei = j + u * G + g;
        int src;
        if (G == 1) {
          src = __builtin_amdgcn_readlane(my_idx, ei & 63);
        } else {
          src = __shfl(my_idx, ei & 63);
        }
        if (CS) s[u] = (G == 1) ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_cs), ei & 63))
                                : __shfl(my_cs, ei & 63);
        const bool ok = (ei < cnt) && col_ok;
        srcs[u] = ok ? src : -1;
        v[u] = ok ? ld4(x + (int64_t)src * ldx + col4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (XF) {
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (srcs[u] >= 0) v[u] = xf_apply(xf, v[u], (uint32_t)srcs[u]);      // (absent edges stay exact zeros)
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc = CS ? fma4(s[u], v[u], acc) : add4(acc, v[u]);
    }
  }
  return acc;
}
template <int LPR, int U, bool CS, bool XF = false>
__device__ __forceinline__ float4 wave_gather_sum(const int32_t* __restrict__ indices, int64_t e0, int64_t e1,
                                                  int wave_id, int n_waves, const float* __restrict__ x,
                                                  int64_t ldx, int col4, bool col_ok,
                                                  const float* __restrict__ col_scale, int lane, const XfCols& xf) {
  return fold_groups<LPR>(wave_gather_acc<LPR, U, CS, XF>(indices, e0, e1, wave_id, n_waves, x, ldx, col4, col_ok, col_scale, lane, xf,
                                                           make_float4(0.f, 0.f, 0.f, 0.f)));
}

// per-column epilogue constants of a lane's four columns, fetched ONCE per wave: inside finish_row they were eight 4-byte loads
// per output row that the compiler could not hoist past the stores
struct EpCols { float s[4], h[4]; };
__device__ __forceinline__ EpCols load_ep_cols(const SpmmArgs& a, int col4) {
  EpCols e;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const bool ok = col4 + t < a.d;
    e.s[t] = (a.ep_scale && ok) ? a.ep_scale[col4 + t] : 1.f;
    e.h[t] = (a.ep_shift && ok) ? a.ep_shift[col4 + t] : 0.f;
  }
  return e;
}

// finish_row with the self row (SAGE_GCN; behind the source transform) already in hand: `s`
template <int MODE>
__device__ __forceinline__ void finish_row_s(const SpmmArgs& a, int64_t v, int64_t deg, float4 acc, float4 s, int col4, const EpCols& ep,
                                             int64_t out_shift = 0) {
  float4 y;
  if (MODE == GLNN_AGG_SAGE_GCN) {
    const float dp1 = (float)deg + 1.0f;
    y = mean4(acc, s, dp1);
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
      yy[t] = 0.f;  // padding columns are written as zero
    }
  }
  st4_stream(a.out + (v + out_shift) * a.ldo + col4, make_float4(yy[0], yy[1], yy[2], yy[3]));
}
template <int MODE, bool XF = false>
__device__ __forceinline__ void finish_row(const SpmmArgs& a, int64_t v, int64_t deg, float4 acc, int col4, const EpCols& ep,
                                           const XfCols& xf, int64_t out_shift = 0, int64_t self_shift = 0) {
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (MODE == GLNN_AGG_SAGE_GCN) {
    const int64_t sr = a.self_rows ? a.self_rows[v] : v + self_shift;
    s = ld4(a.x_self + sr * a.ld_self + col4);
    if (XF) s = xf_apply(xf, s, (uint32_t)sr);
  }
  finish_row_s<MODE>(a, v, deg, acc, s, col4, ep, out_shift);
}

// ---- hub rows ---------------------------------------------------------------------------------------------------------
// index of row v in the plan (uniform), -1 if absent
__device__ __forceinline__ int hub_find(const HubPlan& h, int64_t v) {
  int lo = 0, hi = h.n_hub - 1;
  while (lo <= hi) {
    const int mid = (lo + hi) >> 1;
    const int64_t r = h.rows[mid];
    if (r == v) return mid;
    if (r < v) lo = mid + 1; else hi = mid - 1;
  }
  return -1;
}
// THIS WAVE's share T_w of a hub row's sum (valid in lanes < LPR): the partial sums P(s, w) of its 64-edge piece of every segment s, each
// gathered into a fresh accumulator, added in ascending s -- read from the plan's slab when hub_gather_kernel left them there, gathered
// here otherwise.  The caller folds the eight shares exactly as it folds the eight wave partials of any long row.  No barrier.
template <int LPR, int U, bool CS, bool XF>
__device__ __forceinline__ float4 hub_wave_share(const HubPlan& h, const int32_t* __restrict__ indices, int64_t v, int64_t e0, int64_t e1,
                                                 const float* __restrict__ x, int64_t ldx, int col4, bool col_ok, const float* __restrict__ col_scale,
                                                 int wave, int lane, const XfCols& xf) {
  float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
  const int hi = h.n_hub > 0 ? hub_find(h, v) : -1;
  if (hi >= 0) {
    if (lane < LPR && col_ok) {
      const int s0 = h.seg_ptr[hi], s1 = h.seg_ptr[hi + 1];
      for (int sg = s0; sg < s1; ++sg) t = add4(t, ld4(h.slab + ((int64_t)sg * 8 + wave) * h.ld_slab + col4));
    }
    return t;
  }
  for (int64_t b = e0; b < e1; b += kHubSeg) {
    const int64_t be = b + kHubSeg < e1 ? b + kHubSeg : e1;
    t = add4(t, wave_gather_sum<LPR, U, CS, XF>(indices, b, be, wave, 8, x, ldx, col4, col_ok, col_scale, lane, xf));
  }
  return t;
}

// one workgroup per segment of a hub row: its eight wave partials -> slab[8 segment + wave]
template <int LPR, int U, bool CS, bool XF>
__global__ __launch_bounds__(kBlock) void hub_gather_kernel(const SpmmArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col4 = (lane % LPR) * 4;
  const bool col_ok = col4 < a.d;
  XfCols xf = {};
  if (XF) xf = load_xf_cols(a, col_ok ? col4 : 0);
  const int sg = (int)blockIdx.x;
  int lo = 0, hi = a.hub.n_hub - 1;                      // the hub row whose segment range holds sg: last h with seg_ptr[h] <= sg
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (a.hub.seg_ptr[mid] <= sg) lo = mid; else hi = mid - 1;
  }
  const int64_t v = a.hub.rows[lo];
  const int64_t e0 = a.indptr[v] + (int64_t)(sg - a.hub.seg_ptr[lo]) * kHubSeg;
  const int64_t e1r = a.indptr[v + 1];
  const int64_t e1 = e0 + kHubSeg < e1r ? e0 + kHubSeg : e1r;
  const float4 acc = wave_gather_sum<LPR, U, CS, XF>(a.indices, e0, e1, wave, 8, a.x, a.ldx, col4, col_ok, a.col_scale, lane, xf);
  if (lane < LPR && col_ok) st4(a.hub.slab + ((int64_t)sg * 8 + wave) * a.hub.ld_slab + col4, acc);
}

// ---- long-row role: scan a strided share of the rows, whole workgroup per long row (deterministic LDS fold) ----
template <int LPR, int U, int MODE, bool CS, bool XF = false>
__device__ __forceinline__ void long_rows_role(const SpmmArgs& a, int lane, int wave, int col4, bool col_ok, const EpCols& ep,
                                               const XfCols& xf) {
  __shared__ int64_t s_rows[kBlock];
  __shared__ int s_count;
  __shared__ float4 s_part[kWavesPerBlock][64];
  const int64_t n_chunks = (a.n_dst + kBlock - 1) / kBlock;
  for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += a.n_long_blocks) {
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
    // chunk c owns the rows congruent to c modulo n_chunks (not a contiguous range): a degree-SORTED node order -- all
    // hub rows at the front -- is dealt round-robin over the chunks and so over the long-row workgroups (a contiguous
    // split put thousands of hub rows into a few workgroups: the D=47 layer ran 10.9 instead of 5.2 ms on such a graph)
    const int64_t r = (int64_t)threadIdx.x * n_chunks + chunk;
    if (r < a.n_dst && (a.indptr[r + 1] - a.indptr[r]) > kLongRow) {
      const int slot = atomicAdd(&s_count, 1);
      s_rows[slot] = r;
    }
    __syncthreads();
    const int n_found = s_count;
    for (int i = 0; i < n_found; ++i) {
      const int64_t v = s_rows[i];
      const int64_t e0 = a.indptr[v], e1 = a.indptr[v + 1];
      float4 acc;
      if (e1 - e0 > kHubRow)        // (uniform: every wave reads the same row)
        acc = hub_wave_share<LPR, U, CS, XF>(a.hub, a.indices, v, e0, e1, a.x, a.ldx, col4, col_ok, a.col_scale, wave, lane, xf);
      else
        acc = wave_gather_sum<LPR, U, CS, XF>(a.indices, e0, e1, wave, kWavesPerBlock, a.x, a.ldx, col4, col_ok, a.col_scale, lane, xf);
      if (lane < LPR) s_part[wave][lane] = acc;
      __syncthreads();
      if (wave == 0 && lane < LPR && col_ok) {
        float4 t = s_part[0][lane];
#pragma unroll
        for (int w = 1; w < kWavesPerBlock; ++w) t = add4(t, s_part[w][lane]);
        finish_row<MODE, XF>(a, v, e1 - e0, t, col4, ep, xf);
      }
      __syncthreads();
    }
  }
}

// The long-row role of a CHUNKED launch (a.cm.n > 0): the same scan and the same per-row sums, walked chunk by chunk -- inside chunk c (rows
// [rs, re)) workgroup b takes the sub-ranges b, b + n_long_blocks, .. of ceil((re - rs) / 512), dealt round-robin over the rows as above --
// and when its share of a chunk is stored the workgroup arrives on the chunk's counter (every long-role workgroup arrives once per non-empty
// chunk: expected = row workgroups of the chunk + n_long_blocks).
template <int LPR, int U, int MODE, bool CS, bool XF>
__device__ __forceinline__ void long_rows_role_chunks(const SpmmArgs& a, const ChunkMap& cm, int lane, int wave, int col4, bool col_ok,
                                                      const EpCols& ep, const XfCols& xf) {
  __shared__ int64_t s_rows[kBlock];
  __shared__ int s_count;
  __shared__ float4 s_part[kWavesPerBlock][64];
  int64_t dealt = 0;
#pragma unroll 1
  for (int c = 0; c < cm.n; ++c) {
    const int64_t rs = (int64_t)cm_tile_start(cm, c) * 32;
    int64_t re = (int64_t)cm_tile_start(cm, c + 1) * 32;
    if (re > a.n_dst) re = a.n_dst;
    if (re <= rs) continue;                               // an empty chunk is never signalled
    const int64_t out_shift = cm_pick(cm.out_shift, c), self_shift = cm_pick(cm.self_shift, c);
    const int64_t n_sub = (re - rs + kBlock - 1) / kBlock;
    // the sub-ranges are dealt on round-robin ACROSS the chunks (a chunk of fewer sub-ranges than long-role workgroups would otherwise keep
    // the same low-numbered workgroups busy chunk after chunk)
    const int64_t first = ((int64_t)blockIdx.x - dealt % a.n_long_blocks + a.n_long_blocks) % a.n_long_blocks;
    dealt += n_sub;
#pragma unroll 1
    for (int64_t sub = first; sub < n_sub; sub += a.n_long_blocks) {
      if (threadIdx.x == 0) s_count = 0;
      __syncthreads();
      const int64_t r = rs + (int64_t)threadIdx.x * n_sub + sub;
      if (r < re && (a.indptr[r + 1] - a.indptr[r]) > kLongRow) {
        const int slot = atomicAdd(&s_count, 1);
        s_rows[slot] = r;
      }
      __syncthreads();
      const int n_found = s_count;
      for (int i = 0; i < n_found; ++i) {
        const int64_t v = s_rows[i];
        const int64_t e0 = a.indptr[v], e1 = a.indptr[v + 1];
        float4 acc;
        if (e1 - e0 > kHubRow)
          acc = hub_wave_share<LPR, U, CS, XF>(a.hub, a.indices, v, e0, e1, a.x, a.ldx, col4, col_ok, a.col_scale, wave, lane, xf);
        else
          acc = wave_gather_sum<LPR, U, CS, XF>(a.indices, e0, e1, wave, kWavesPerBlock, a.x, a.ldx, col4, col_ok, a.col_scale, lane, xf);
        if (lane < LPR) s_part[wave][lane] = acc;
        __syncthreads();
        if (wave == 0 && lane < LPR && col_ok) {
          float4 t = s_part[0][lane];
#pragma unroll
          for (int w = 1; w < kWavesPerBlock; ++w) t = add4(t, s_part[w][lane]);
          finish_row<MODE, XF>(a, v, e1 - e0, t, col4, ep, xf, out_shift, self_shift);
        }
        __syncthreads();
      }
    }
    // this workgroup's share of chunk c is stored (only wave 0 stores): acknowledged, then one arrival
    if (wave == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) {
        const int64_t rpb = a.rows_per_block;
        cm_arrive(cm, c, (int)((re - rs + rpb - 1) / rpb) + a.n_long_blocks);
      }
    }
  }
}

template <int LPR, int U, int MODE, bool CS, bool XF = false, bool CM = false>
__global__ __launch_bounds__(kBlock) void spmm_csr_kernel(const SpmmArgs a0) {
  // rows wider than 256 floats (raw cora / citeseer features): blockIdx.y = the 256-column tile of this workgroup -- one launch instead
  // of one per tile (14 for citeseer's 3703 features)
  SpmmArgs a = a0;
  if (gridDim.y > 1) {
    const int off = 256 * (int)blockIdx.y;
    a.x += off; a.out += off; a.d = a0.d - off < 256 ? a0.d - off : 256;
    if (a.x_self) a.x_self += off;
    if (a.ep_scale) a.ep_scale += off;
    if (a.ep_shift) a.ep_shift += off;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane % LPR;
  const int col4 = c * 4;
  const bool col_ok = col4 < a.d;
  const EpCols ep = load_ep_cols(a, col_ok ? col4 : 0);
  XfCols xf = {};                                     // (registers; dead code when !XF)
  if (XF) xf = load_xf_cols(a, col_ok ? col4 : 0);

  if ((int)blockIdx.x < a.n_long_blocks) {
    if constexpr (CM) long_rows_role_chunks<LPR, U, MODE, CS, XF>(a, a0.cm, lane, wave, col4, col_ok, ep, xf);
    else long_rows_role<LPR, U, MODE, CS, XF>(a, lane, wave, col4, col_ok, ep, xf);
    return;
  }

  // ---- row role: one wave per row; the 8 waves of the workgroup pull its 8*kRowsPerWave rows from an LDS ticket
  //      (degrees vary by 100x: a static split leaves waves idle behind the heaviest one; measured -8 % at D=256) ----
  __shared__ int s_ticket;
  __shared__ int s_arrived;                            // (CM: waves of this workgroup that are done)
  if (threadIdx.x == 0) { s_ticket = 0; if (CM) s_arrived = 0; }
  __syncthreads();
  const int64_t blk = (int64_t)blockIdx.x - a.n_long_blocks;
  const int64_t row_base = blk * a.rows_per_block;
  int chunk = 0;
  int64_t out_shift = 0, self_shift = 0;
  if constexpr (CM) {                                  // (chunk boundaries are multiples of rows_per_block: the workgroup lies inside one chunk)
#pragma unroll
    for (int c = 1; c < GLNN_MAX_CHUNKS; ++c)
      if (c < a0.cm.n && row_base >= (int64_t)a0.cm.tile_start[c] * 32) chunk = c;
    out_shift = cm_pick(a0.cm.out_shift, chunk);
    self_shift = cm_pick(a0.cm.self_shift, chunk);
  }
#pragma unroll 1
  while (true) {
    int lr = 0;
    if (lane == 0) lr = atomicAdd(&s_ticket, 1);
    lr = __builtin_amdgcn_readfirstlane(lr);
    if (lr >= a.rows_per_block) break;
    const int64_t v = row_base + lr;
    if (v >= a.n_dst) break;
    const int64_t e0 = a.indptr[v], e1 = a.indptr[v + 1];
    const int64_t deg = e1 - e0;
    if (deg > kLongRow) continue;
    float4 acc = wave_gather_sum<LPR, U, CS, XF>(a.indices, e0, e1, 0, 1, a.x, a.ldx, col4, col_ok, a.col_scale, lane, xf);
    if (lane < LPR && col_ok) finish_row<MODE, XF>(a, v, deg, acc, col4, ep, xf, out_shift, self_shift);
  }
  if constexpr (CM) {
    // stores acknowledged -> the eight waves count themselves in LDS -> the last one arrives for the workgroup (see chunk_arrive)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && atomicAdd(&s_arrived, 1) == kWavesPerBlock - 1) {
      const int64_t rs = (int64_t)cm_tile_start(a0.cm, chunk) * 32;
      int64_t re = (int64_t)cm_tile_start(a0.cm, chunk + 1) * 32;
      if (re > a.n_dst) re = a.n_dst;
      const int64_t rpb = a.rows_per_block;
      cm_arrive(a0.cm, chunk, (int)((re - rs + rpb - 1) / rpb) + a.n_long_blocks);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// SHORT ROWS (round 5): the outermost block of a sampled training batch (fan-out 5) has rows whose whole gather is ONE batch of loads.
// One wave per row then spends its time in the row's dependent chain (ticket -> indptr -> indices -> rows -> store, ~4 memory latencies
// for 2.4 KB moved): 0.5 M rows x 6 x 400 bytes in 365 us = 3.8 TB/s.  Here a wave takes FOUR consecutive rows per ticket: one load for their five indptr entries (and self-row
// ids), one for their (contiguous) column indices, then the first three load units of all four rows and the four self rows in flight
// together; longer rows continue on their own with the running sums carried on.  A group of lanes sees a row's edges in the same order and
// the same fold as in spmm_csr_kernel, so the two kernels give the same bits; the long-row role is shared.
// ---------------------------------------------------------------------------------------------
constexpr int kShortRows = 4;        // rows per ticket
constexpr int kShortUnits = 3;       // load units (G edges each) of every row gathered with the batch
__device__ __forceinline__ int64_t readlane64(int64_t v, int l) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), l);
  return (int64_t)(((uint64_t)hi << 32) | lo);
}
template <int LPR, int U, int MODE, bool CS, bool XF = false>
__global__ __launch_bounds__(kBlock) void spmm_csr_short_kernel(const SpmmArgs a) {
  constexpr int G = 64 / LPR, RB = kShortRows, UB = kShortUnits;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane % LPR, g = lane / LPR;
  const int col4 = c * 4;
  const bool col_ok = col4 < a.d;
  const EpCols ep = load_ep_cols(a, col_ok ? col4 : 0);
  XfCols xf = {};
  if (XF) xf = load_xf_cols(a, col_ok ? col4 : 0);
  if ((int)blockIdx.x < a.n_long_blocks) {
    long_rows_role<LPR, U, MODE, CS, XF>(a, lane, wave, col4, col_ok, ep, xf);
    return;
  }
  __shared__ int s_ticket;
  if (threadIdx.x == 0) s_ticket = 0;
  __syncthreads();
  const int64_t blk = (int64_t)blockIdx.x - a.n_long_blocks;
  const int64_t row_base = blk * a.rows_per_block;
#pragma unroll 1
  while (true) {
    int lr = 0;
    if (lane == 0) lr = atomicAdd(&s_ticket, RB);
    lr = __builtin_amdgcn_readfirstlane(lr);
    if (lr >= a.rows_per_block) break;
    const int64_t v0 = row_base + lr;
    if (v0 >= a.n_dst) break;
    int nr = a.rows_per_block - lr;
    if (nr > RB) nr = RB;
    if (v0 + nr > a.n_dst) nr = (int)(a.n_dst - v0);
    // lanes 0 .. nr: the rows' indptr entries (and the self-row ids of global-id blocks) in one request each
    const int64_t my_ptr = a.indptr[v0 + (lane <= nr ? lane : nr)];
    int64_t my_self = v0 + (lane < nr ? lane : 0);
    if (MODE == GLNN_AGG_SAGE_GCN && a.self_rows) my_self = a.self_rows[v0 + (lane < nr ? lane : 0)];
    int64_t e[RB + 1];
#pragma unroll
    for (int r = 0; r <= RB; ++r) e[r] = readlane64(my_ptr, r <= nr ? r : nr);
    const int64_t ebase = e[0];
    int64_t span = e[RB] - ebase;
    const int tot = span < 64 ? (int)span : 64;
    const int my_idx = lane < tot ? ld_idx_stream(a.indices + ebase + lane) : 0;
    float my_cs = 0.f;
    if (CS) my_cs = lane < tot ? a.col_scale[my_idx] : 0.f;
    int hd[RB];                                            // edges of row r gathered with the batch (a multiple of G, or the whole row)
    bool live[RB];                                         // a row of this wave (rows above kLongRow belong to the long-row role)
    float4 v[RB][UB], selfv[RB];
    float sc[RB][UB];
    int srcs[RB][UB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int64_t deg = e[r + 1] - e[r];
      const int off = (int)(e[r] - ebase < 64 ? e[r] - ebase : 64);
      live[r] = r < nr && deg <= kLongRow;
      int h = deg < UB * G ? (int)deg : UB * G;
      if (!live[r] || off + h > tot) h = 0;                // (its first units do not lie inside the loaded indices: the row goes alone)
      hd[r] = h;
#pragma unroll
      for (int k = 0; k < UB; ++k) {
        const int first = k * G + g;
        const int ei = off + first;
        int src;
        if (G == 1) src = __builtin_amdgcn_readlane(my_idx, ei & 63);
        else src = __shfl(my_idx, ei & 63);
        float cs = 0.f;
        if (CS) cs = (G == 1) ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_cs), ei & 63)) : __shfl(my_cs, ei & 63);
        const bool ok = first < h && col_ok;
        srcs[r][k] = ok ? src : -1;
        sc[r][k] = ok ? cs : 0.f;
        v[r][k] = ok ? ld4(a.x + (int64_t)src * a.ldx + col4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      selfv[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (MODE == GLNN_AGG_SAGE_GCN) {
        const int64_t sr = readlane64(my_self, r < nr ? r : 0);
        if (live[r] && col_ok) {
          selfv[r] = ld4(a.x_self + sr * a.ld_self + col4);
          if (XF) selfv[r] = xf_apply(xf, selfv[r], (uint32_t)sr);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < UB; ++k) {
        if (XF) {
          if (srcs[r][k] >= 0) v[r][k] = xf_apply(xf, v[r][k], (uint32_t)srcs[r][k]);
        }
        acc = CS ? fma4(sc[r][k], v[r][k], acc) : add4(acc, v[r][k]);
      }
      if (!live[r]) continue;                              // (uniform)
      const int64_t deg = e[r + 1] - e[r];
      if (deg > hd[r])                                     // the rest of a longer row, its group sums carried on
        acc = wave_gather_acc<LPR, U, CS, XF>(a.indices, e[r] + hd[r], e[r + 1], 0, 1, a.x, a.ldx, col4, col_ok, a.col_scale, lane, xf, acc);
      acc = fold_groups<LPR>(acc);
      if (lane < LPR && col_ok) finish_row_s<MODE>(a, v0 + r, deg, acc, selfv[r], col4, ep);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// A^T dY WITH the first pass of the BatchNorm backward behind it (glnn::spmm_csr_bn_dy, round 5): the transposed aggregation of a training
// step produces da = dL/dh of a hidden layer whose tail is BatchNorm -> ReLU -> dropout.  Its BatchNorm backward starts with a pass over
// (da, z) for dy = da behind the masks and the column sums S1 = sum dy, S2 = sum dy xhat.  Here the aggregation's epilogue does that pass
// on the row it holds: it reads the row of z, stores dy INSTEAD of da and accumulates the row into per-wave column sums -- da is never
// written or read back (0.5 M x 256 on the products configuration: 1 GB of traffic and a launch).
// Deterministic by construction: rows are dealt to the waves STATICALLY (row_base + wave + 8 i -- no ticket), a wave adds its rows in
// ascending order, the eight waves of a workgroup are folded in fixed order into the workgroup's slot [blockIdx.x][h] of ws1 / ws2; the
// long-row workgroups take their rows in ascending id order.  glnn::bn_bwd_deferred_finish folds the slots.
// ---------------------------------------------------------------------------------------------
struct DyTail {
  const float* z; int64_t ldz; const float* mean; const float* rstd; const float* a_scale; const float* a_shift;
  uint32_t dthr; uint32_t dseed; float dscale; int relu; float* ws1; float* ws2; int h;
};
struct DyCols { float mu[4], rs[4], sc[4], sf[4]; };
__device__ __forceinline__ DyCols load_dy_cols(const DyTail& t, int col4) {
  DyCols c;
  const float4 mu = ld4(t.mean + col4), rs = ld4(t.rstd + col4), sc = ld4(t.a_scale + col4), sf = ld4(t.a_shift + col4);
  c.mu[0] = mu.x; c.mu[1] = mu.y; c.mu[2] = mu.z; c.mu[3] = mu.w;
  c.rs[0] = rs.x; c.rs[1] = rs.y; c.rs[2] = rs.z; c.rs[3] = rs.w;
  c.sc[0] = sc.x; c.sc[1] = sc.y; c.sc[2] = sc.z; c.sc[3] = sc.w;
  c.sf[0] = sf.x; c.sf[1] = sf.y; c.sf[2] = sf.z; c.sf[3] = sf.w;
  return c;
}
// one finished row: da (the aggregate) -> dy, stored; the row added to the caller's running column sums (bn_bwd_partial's expressions)
template <bool DROP>
__device__ __forceinline__ void finish_dy_row(const SpmmArgs& a, const DyTail& t, const DyCols& c, int64_t v, float4 acc, float4 z4, int col4,
                                              float (&s1)[4], float (&s2)[4]) {
  const float da[4] = {acc.x, acc.y, acc.z, acc.w}, zz[4] = {z4.x, z4.y, z4.z, z4.w};
  float o[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float dav = da[k];
    if (DROP) dav = glnn::drop_keep(t.dseed, t.dthr, (uint32_t)v, (uint32_t)(col4 + k)) ? dav * t.dscale : 0.f;
    const float dy = (!t.relu || fmaf(zz[k], c.sc[k], c.sf[k]) > 0.f) ? dav : 0.f;
    s1[k] += dy;
    s2[k] = fmaf(dy, (zz[k] - c.mu[k]) * c.rs[k], s2[k]);
    o[k] = dy;
  }
  st4_stream(a.out + v * a.ldo + col4, make_float4(o[0], o[1], o[2], o[3]));
}

#ifndef GLNN_BN_DY_BATCH
#define GLNN_BN_DY_BATCH 1
#endif
template <int LPR, int U, bool DROP>
__global__ __launch_bounds__(kBlock) void spmm_bn_dy_kernel(const SpmmArgs a, const DyTail t) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col4 = (lane % LPR) * 4;
  const bool col_ok = col4 < a.d;                        // d % 4 == 0: a lane's four columns are all inside or all outside
  const DyCols c = load_dy_cols(t, col_ok ? col4 : 0);
  const XfCols xf = {};
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  __shared__ float4 s_part[kWavesPerBlock][64], s_part2[kWavesPerBlock][64];
  if ((int)blockIdx.x < a.n_long_blocks) {
    // ---- long rows: the scan of long_rows_role, the found rows of a chunk taken in ASCENDING order (the sums must not depend on the
    //      order in which the lanes' atomics landed) ----
    __shared__ int64_t s_rows[kBlock], s_sorted[kBlock];
    __shared__ int s_count;
    const int64_t n_chunks = (a.n_dst + kBlock - 1) / kBlock;
    for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += a.n_long_blocks) {
      if (threadIdx.x == 0) s_count = 0;
      __syncthreads();
      const int64_t r = (int64_t)threadIdx.x * n_chunks + chunk;
      if (r < a.n_dst && (a.indptr[r + 1] - a.indptr[r]) > kLongRow) s_rows[atomicAdd(&s_count, 1)] = r;
      __syncthreads();
      const int n_found = s_count;
      if ((int)threadIdx.x < n_found) {
        const int64_t mine = s_rows[threadIdx.x];
        int rank = 0;
        for (int i = 0; i < n_found; ++i) rank += s_rows[i] < mine ? 1 : 0;
        s_sorted[rank] = mine;
      }
      __syncthreads();
      for (int i = 0; i < n_found; ++i) {
        const int64_t v = s_sorted[i];
        const int64_t e0 = a.indptr[v], e1 = a.indptr[v + 1];
        float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (wave == 0 && lane < LPR && col_ok) z4 = ld4(t.z + v * t.ldz + col4);
        float4 acc;
        if (e1 - e0 > kHubRow) acc = hub_wave_share<LPR, U, true, false>(a.hub, a.indices, v, e0, e1, a.x, a.ldx, col4, col_ok, a.col_scale, wave, lane, xf);
        else acc = wave_gather_sum<LPR, U, true, false>(a.indices, e0, e1, wave, kWavesPerBlock, a.x, a.ldx, col4, col_ok, a.col_scale, lane, xf);
        if (lane < LPR) s_part[wave][lane] = acc;
        __syncthreads();
        if (wave == 0 && lane < LPR && col_ok) {
          float4 sum = s_part[0][lane];
#pragma unroll
          for (int w = 1; w < kWavesPerBlock; ++w) sum = add4(sum, s_part[w][lane]);
          finish_dy_row<DROP>(a, t, c, v, sum, z4, col4, s1, s2);
        }
        __syncthreads();
      }
    }
    if (wave == 0 && lane < LPR && col_ok) {             // the long-row workgroup's slot: wave 0's sums
      st4(t.ws1 + (int64_t)blockIdx.x * t.h + col4, make_float4(s1[0], s1[1], s1[2], s1[3]));
      st4(t.ws2 + (int64_t)blockIdx.x * t.h + col4, make_float4(s2[0], s2[1], s2[2], s2[3]));
    }
    return;
  }
  // ---- row role, static: wave w takes the rows row_base + w + 8 i ----
  const int64_t blk = (int64_t)blockIdx.x - a.n_long_blocks;
  const int64_t row_base = blk * a.rows_per_block;
#if GLNN_BN_DY_BATCH
  // ... kShortRows of ITS rows at a time (the transposed blocks of a sampled batch have 1-2 entries per row: a wave that took one row after
  // the other spent its time in the row's chain indptr -> index -> column scale / source row -> store, at 2.9 TB/s of a launch that streams
  // z in and dy out): the indptr pairs of the four rows in one request, their first two load units, column scales and z rows in
  // flight together; a longer row continues on its own with its sums carried on.  The rows are finished in ascending order and a row's
  // units are added in ascending order, so dy AND the column sums are the bits of the one-row-at-a-time form (GLNN_BN_DY_BATCH=0).
  {
    constexpr int G = 64 / LPR, RB = kShortRows, UB = 2;      // (two units: 1-2 entries per row, and the batch stays inside 128 VGPRs -- two workgroups per CU)
    const int g = lane / LPR;
#pragma unroll 1
    for (int lr0 = wave; lr0 < a.rows_per_block; lr0 += kWavesPerBlock * RB) {
      int nr = 0;                                            // this wave's rows of the batch: a prefix (rows ascend)
#pragma unroll
      for (int r = 0; r < RB; ++r)
        if (lr0 + kWavesPerBlock * r < a.rows_per_block && row_base + lr0 + kWavesPerBlock * r < a.n_dst) nr = r + 1;
      if (nr == 0) break;
      const int rl = (lane >> 1) < nr ? (lane >> 1) : 0;     // lanes 2r, 2r + 1: indptr[v_r], indptr[v_r + 1]
      const int64_t my_ptr = a.indptr[row_base + lr0 + kWavesPerBlock * rl + (lane & 1)];
      int64_t e0[RB], e1[RB];
      bool live[RB];
      int hd[RB], my_idx[RB];
      float my_cs[RB];
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        e0[r] = readlane64(my_ptr, 2 * r);
        e1[r] = readlane64(my_ptr, 2 * r + 1);
        const int64_t deg = e1[r] - e0[r];
        live[r] = r < nr && deg <= kLongRow;                 // (rows above kLongRow belong to the long-row role)
        hd[r] = live[r] ? (deg < UB * G ? (int)deg : UB * G) : 0;
        my_idx[r] = lane < hd[r] ? ld_idx_stream(a.indices + e0[r] + lane) : 0;
      }
#pragma unroll
      for (int r = 0; r < RB; ++r) my_cs[r] = lane < hd[r] ? a.col_scale[my_idx[r]] : 0.f;
      float4 v[RB][UB], z4[RB];
      float sc[RB][UB];
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        z4[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live[r] && lane < LPR && col_ok) z4[r] = ld4(t.z + (row_base + lr0 + kWavesPerBlock * r) * t.ldz + col4);
#pragma unroll
        for (int k = 0; k < UB; ++k) {
          const int first = k * G + g;
          int src;
          float cs;
          if (G == 1) {
            src = __builtin_amdgcn_readlane(my_idx[r], first & 63);
            cs = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_cs[r]), first & 63));
          } else {
            src = __shfl(my_idx[r], first & 63);
            cs = __shfl(my_cs[r], first & 63);
          }
          const bool ok = first < hd[r] && col_ok;
          sc[r][k] = ok ? cs : 0.f;
          v[r][k] = ok ? ld4(a.x + (int64_t)src * a.ldx + col4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        if (!live[r]) continue;                              // (uniform)
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < UB; ++k) acc = fma4(sc[r][k], v[r][k], acc);
        if (e1[r] - e0[r] > hd[r])
          acc = wave_gather_acc<LPR, U, true, false>(a.indices, e0[r] + hd[r], e1[r], 0, 1, a.x, a.ldx, col4, col_ok, a.col_scale, lane, xf, acc);
        acc = fold_groups<LPR>(acc);
        if (lane < LPR && col_ok) finish_dy_row<DROP>(a, t, c, row_base + lr0 + kWavesPerBlock * r, acc, z4[r], col4, s1, s2);
      }
    }
  }
#else
#pragma unroll 1
  for (int lr = wave; lr < a.rows_per_block; lr += kWavesPerBlock) {
    const int64_t v = row_base + lr;
    if (v >= a.n_dst) break;
    const int64_t e0 = a.indptr[v], e1 = a.indptr[v + 1];
    if (e1 - e0 > kLongRow) continue;
    float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < LPR && col_ok) z4 = ld4(t.z + v * t.ldz + col4);          // (in flight beside the gather)
    const float4 acc = wave_gather_sum<LPR, U, true, false>(a.indices, e0, e1, 0, 1, a.x, a.ldx, col4, col_ok, a.col_scale, lane, xf);
    if (lane < LPR && col_ok) finish_dy_row<DROP>(a, t, c, v, acc, z4, col4, s1, s2);
  }
#endif
  if (lane < LPR) {
    s_part[wave][lane] = make_float4(s1[0], s1[1], s1[2], s1[3]);
    s_part2[wave][lane] = make_float4(s2[0], s2[1], s2[2], s2[3]);
  }
  __syncthreads();
  if (wave == 0 && lane < LPR && col_ok) {
    float4 u = s_part[0][lane], w2 = s_part2[0][lane];
#pragma unroll
    for (int w = 1; w < kWavesPerBlock; ++w) { u = add4(u, s_part[w][lane]); w2 = add4(w2, s_part2[w][lane]); }
    st4(t.ws1 + (int64_t)blockIdx.x * t.h + col4, u);
    st4(t.ws2 + (int64_t)blockIdx.x * t.h + col4, w2);
  }
}

// ---------------------------------------------------------------------------------------------
// K1F: fused SAGE-"gcn" layer  out = epi( ((sum_{u->v} x[u] + x_self[v]) / (deg+1)) @ W^T )
// for aggregate-first layers (d_in <= 256, d_out <= 256): the aggregated rows never go to HBM.
//   phase A  the 8 waves of a workgroup aggregate a tile of 32 destination rows (4 rows each, the same
//            wave_gather_sum as the stand-alone kernel) and park the normalised rows in LDS;
//            rows of degree > LONG_ROW in the tile are then taken by all 8 waves together;
//   phase B  wave w multiplies the LDS tile [32 x K] with the w-th 32-column panel of W on the fp32 MFMA
//            (v_mfma_f32_32x32x2_f32).  W arrives PRE-PACKED in MFMA B-fragment order (glnn_pack_weight_f32),
//            so each k-group is one coalesced 1 KiB load per wave straight from L2 -- no LDS staging and no
//            barrier inside the GEMM; the A fragments are ds_read_b128 from the padded (conflict-free) tile;
//   epilogue per-column scale/shift (+bias, eval BatchNorm) and ReLU, stored from the MFMA accumulators.
// Workgroups on the same CU run out of phase, so the short MFMA phase of one hides under the HBM-bound
// aggregation of the others.
// ---------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kFusedWaves = 8;                      // one wave per 32-column panel of W (d_out <= 256)
constexpr int kFusedBlock = 64 * kFusedWaves;       // 16-wave workgroups measured 1.5x slower (they drain badly)

// How a chunk's rows become visible to the stream that waits for its signal (three forms were measured, N = 8 emulated, the 256 -> 256 -> 47
// layer of a rank, profiles/r06_one_launch_ab.txt): a release per storing wave (buffer_wbl2 sc0 sc1 = a write-back of the XCD's whole L2,
// 19 k of them per launch) took 3.45 ms instead of 2.55; write-through stores (sc0 sc1) + s_waitcnt cost nothing on the 47-wide rows (2.42 ms)
// but +17 % on a 256-wide output (dword stores are not combined on their way to memory).  Shipped: PLAIN stores, acknowledged by the L2
// (s_waitcnt vmcnt(0)) before the wave arrives; the write-back to memory is ONE per chunk and comes from the waiting side --
// glnn_stream_wait_value32 puts an empty kernel behind the wait, whose end-of-kernel release writes every XCD's L2 back, as between any two
// kernels.
template <bool CM>
__device__ __forceinline__ void store_out(float* p, float v) {
  *p = v;
}

template <bool CM>
__device__ __forceinline__ void chunk_arrive(const ChunkMap& cm, int chunk, bool stored, int lane, int* s_arrived) {
  if constexpr (CM) {
    if (stored) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the wave's stores have been acknowledged (they sit in this XCD's L2)
    if (lane != 0) return;
    // the workgroup's eight waves count themselves in LDS; the last one arrives for the tile (ONE global atomic per tile: with one per
    // wave the 100 -> 256 layer issued a same-address atomic every 16 ns and ran 17 % slower)
    if (atomicAdd(s_arrived, 1) != kFusedWaves - 1) return;
    int* cnt = cm.arrivals + chunk;
    const int prev = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int expected = 0;
    uint32_t* sig = nullptr;
#pragma unroll
    for (int c = 0; c < GLNN_MAX_CHUNKS; ++c)
      if (c == chunk) { expected = cm.tile_start[c + 1] - cm.tile_start[c]; sig = cm.signal[c]; }
    if (prev == expected - 1) {
      // every other tile's rows were acknowledged before its add was performed; nothing of theirs is read here
      __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sig, cm.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

struct FusedArgs {
  const int64_t* indptr; const int32_t* indices; int64_t n_dst;
  const float* x; int64_t ldx; int d_in;
  const float* x_self; int64_t ld_self;
  const float* w_packed; int d_out; int kgroups;      // kgroups = ceil(d_in / 8)
  const float* ep_scale; const float* ep_shift; int relu;
  float* out; int64_t ldo;                              // may be NULL when the chained projection below is the only consumer
  // optional chained projection: out2 = epi(...) @ W2^T for the NEXT layer when that layer projects first (in > out): the
  // hidden rows go from the MFMA accumulators through LDS into a second MFMA pass and never reach HBM
  const float* w2_packed; int d_out2; int kgroups2; float* out2; int64_t ldo2;
  const int32_t* tile_order;                            // optional permutation of the tile ids (heaviest tiles first)
  HubPlan hub;                                          // n_hub == 0: hub rows are summed by the tile's own workgroup
  ChunkMap cm;                                          // n == 0: off (the kernel's CM = false instantiation never reads it)
};

template <int LPR, int U, int RT, bool CM = false>      // RT = 32-row sub-tiles per workgroup: each W fragment load feeds RT MFMA chains; CM: chunk map
__global__ __launch_bounds__(kFusedBlock) void sage_fused_kernel(const FusedArgs a) {
  constexpr int kFusedRows = 32 * RT;
  extern __shared__ __attribute__((aligned(16))) float lds_a[];      // [kFusedRows][kpad + 4]
  __shared__ float4 s_part[kFusedWaves / 2][64];   // 4 KiB: with the 33 KiB tile of K=256 this keeps 4 workgroups per CU
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col4 = (lane % LPR) * 4;
  const bool col_ok = col4 < a.d_in;
  const int kpad = a.kgroups * 8;
  const int lda = kpad + 4;
  const int tile_id = a.tile_order ? a.tile_order[blockIdx.x] : (int)blockIdx.x;
  const int64_t row0 = (int64_t)tile_id * kFusedRows;
  int chunk = 0;
  int64_t self_shift = 0, out_shift = 0;
  if constexpr (CM) {
    self_shift = a.cm.self_shift[0];
    out_shift = a.cm.out_shift[0];
#pragma unroll
    for (int c = 1; c < GLNN_MAX_CHUNKS; ++c)
      if (c < a.cm.n && tile_id >= a.cm.tile_start[c]) { chunk = c; self_shift = a.cm.self_shift[c]; out_shift = a.cm.out_shift[c]; }
  }

  // ---- phase A: the 8 waves pull rows of the tile from an LDS ticket (degrees vary by 100x: a static
  //      4-rows-per-wave split leaves most waves idle at the barrier behind the heaviest one) ---------
  __shared__ int s_next;
  __shared__ int s_arrived;                            // (CM: waves of this workgroup that are done)
  if (threadIdx.x == 0) { s_next = 0; if (CM) s_arrived = 0; }
  __syncthreads();
#pragma unroll 1
  while (true) {
    int lr = 0;
    if (lane == 0) lr = atomicAdd(&s_next, 1);
    lr = __builtin_amdgcn_readfirstlane(lr);
    if (lr >= kFusedRows) break;
    const int64_t v = row0 + lr;
    float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
    bool deferred = false;
    if (v < a.n_dst) {
      const int64_t e0 = a.indptr[v], e1 = a.indptr[v + 1];
      const int64_t deg = e1 - e0;
      deferred = deg > kLongRow;
      if (!deferred) {
        const float4 acc = wave_gather_sum<LPR, U, false>(a.indices, e0, e1, 0, 1, a.x, a.ldx, col4, col_ok, nullptr, lane, XfCols{});
        if (lane < LPR && col_ok) {
          const float4 sf = ld4(a.x_self + (v + self_shift) * a.ld_self + col4);
          const float dp1 = (float)deg + 1.0f;
          y = mean4(acc, sf, dp1);
          if (col4 + 1 >= a.d_in) y.y = 0.f;
          if (col4 + 2 >= a.d_in) y.z = 0.f;
          if (col4 + 3 >= a.d_in) y.w = 0.f;
        }
      }
    }
    if (!deferred && lane < LPR && col4 < kpad) st4(lds_a + lr * lda + col4, y);
  }
  __syncthreads();
  // long rows of this tile: all 8 waves on one row at a time (uniform loop: every wave sees the same degrees)
#pragma unroll 1
  for (int lr = 0; lr < kFusedRows; ++lr) {
    const int64_t v = row0 + lr;
    if (v >= a.n_dst) break;
    const int64_t e0 = a.indptr[v], e1 = a.indptr[v + 1];
    if (e1 - e0 <= kLongRow) continue;
    const float4 acc = e1 - e0 > kHubRow
        ? hub_wave_share<LPR, U, false, false>(a.hub, a.indices, v, e0, e1, a.x, a.ldx, col4, col_ok, nullptr, wave, lane, XfCols{})
        : wave_gather_sum<LPR, U, false>(a.indices, e0, e1, wave, kFusedWaves, a.x, a.ldx, col4, col_ok, nullptr, lane, XfCols{});
    // fold 8 wave partials through 4 LDS slots, fixed order: waves 4-7 park, waves 0-3 add theirs, wave 0 sums
    if (wave >= 4 && lane < LPR) s_part[wave - 4][lane] = acc;
    __syncthreads();
    if (wave < 4 && lane < LPR) s_part[wave][lane] = add4(acc, s_part[wave][lane]);
    __syncthreads();
    if (wave == 0 && lane < LPR && col4 < kpad) {
      const float4 t = add4(add4(s_part[0][lane], s_part[1][lane]), add4(s_part[2][lane], s_part[3][lane]));
      float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
      if (col_ok) {
        const float4 sf = ld4(a.x_self + (v + self_shift) * a.ld_self + col4);
        const float dp1 = (float)(e1 - e0) + 1.0f;
        y = mean4(t, sf, dp1);
        if (col4 + 1 >= a.d_in) y.y = 0.f;
        if (col4 + 2 >= a.d_in) y.z = 0.f;
        if (col4 + 3 >= a.d_in) y.w = 0.f;
      }
      st4(lds_a + lr * lda + col4, y);
    }
    __syncthreads();
  }

  // ---- phase B: [32*RT x K] (LDS) x W panel `wave` (packed, L2) on the MFMA -------------------------
  const int n_tiles = (a.d_out + 31) / 32;
  const int nt = wave;                                 // column panel of W
  const bool chain = a.w2_packed != nullptr;
  if (nt >= n_tiles && !chain) { chunk_arrive<CM>(a.cm, chunk, false, lane, &s_arrived); return; }
  const int li = lane & 31, kk = lane >> 5;
  f32x16 acc[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  if (nt < n_tiles) {
    const float4* wp = reinterpret_cast<const float4*>(a.w_packed) + ((int64_t)nt * a.kgroups) * 64 + lane;
    const float* ap = lds_a + li * lda + kk * 4;
    constexpr int PF = 4;                      // B fragments in flight
    float4 bq[PF];
#pragma unroll
    for (int q = 0; q < PF; ++q) bq[q] = (q < a.kgroups) ? wp[(int64_t)q * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int kg0 = 0; kg0 < a.kgroups; kg0 += PF) {
#pragma unroll
      for (int q = 0; q < PF; ++q) {
        const int kg = kg0 + q;
        if (kg < a.kgroups) {
          const float4 bv = bq[q];
          const int nxt = kg + PF;
          if (nxt < a.kgroups) bq[q] = wp[(int64_t)nxt * 64];
          float4 av[RT];
#pragma unroll
          for (int t = 0; t < RT; ++t) av[t] = *reinterpret_cast<const float4*>(ap + t * 32 * lda + kg * 8);
#pragma unroll
          for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t].x, bv.x, acc[t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t].y, bv.y, acc[t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t].z, bv.z, acc[t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t].w, bv.w, acc[t], 0, 0, 0);
        }
      }
    }
  }
  // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) -----
  const int col = nt * 32 + li;
  const int ldh = a.kgroups2 * 8 + 4;                  // row stride of the hidden tile parked in LDS for the chained pass
  if (chain) __syncthreads();                          // every wave is done reading the aggregate tile: the LDS is reused
  if (nt < n_tiles) {
    const bool col_ok2 = col < a.d_out;
    const float es = (a.ep_scale && col_ok2) ? a.ep_scale[col] : 1.f;
    const float eh = (a.ep_shift && col_ok2) ? a.ep_shift[col] : 0.f;
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lr = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        const int64_t row = row0 + lr;
        float v = fmaf(acc[t][r], es, eh);
        if (a.relu) v = fmaxf(v, 0.f);
        if (!col_ok2) v = 0.f;
        if (a.out && col_ok2 && row < a.n_dst) store_out<CM>(a.out + (row + out_shift) * a.ldo + col, v);
        if (chain && col < a.kgroups2 * 8) lds_a[lr * ldh + col] = v;
      }
  }
  if (!chain) { chunk_arrive<CM>(a.cm, chunk, true, lane, &s_arrived); return; }
  __syncthreads();
  // ---- phase C: [32*RT x d_out] hidden tile (LDS) x W2 panel `wave` on the MFMA -> out2 ----------------
  const int n_tiles2 = (a.d_out2 + 31) / 32;
  if (nt >= n_tiles2) { chunk_arrive<CM>(a.cm, chunk, a.out != nullptr && nt < n_tiles, lane, &s_arrived); return; }
  {
    const float4* wp = reinterpret_cast<const float4*>(a.w2_packed) + ((int64_t)nt * a.kgroups2) * 64 + lane;
    const float* ap = lds_a + li * ldh + kk * 4;
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int kg = 0; kg < a.kgroups2; ++kg) {
      const float4 bv = wp[(int64_t)kg * 64];
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const float4 av = *reinterpret_cast<const float4*>(ap + t * 32 * ldh + kg * 8);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[t], 0, 0, 0);
      }
    }
    const int col2 = nt * 32 + li;
    if (col2 < a.d_out2) {
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = row0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
          if (row < a.n_dst) store_out<CM>(a.out2 + (row + out_shift) * a.ldo2 + col2, acc[t][r]);
        }
    }
  }
  chunk_arrive<CM>(a.cm, chunk, true, lane, &s_arrived);
}

// W [d_out, d_in] (ldw) -> MFMA B-fragment order: wp[nt][kg][lane][t] = W[nt*32 + (lane&31)][kg*8 + (lane>>5)*4 + t]
__global__ void pack_weight_kernel(const float* __restrict__ w, int64_t ldw, int d_out, int d_in, int kgroups,
                                   float* __restrict__ wp) {
  const int64_t total = (int64_t)((d_out + 31) / 32) * kgroups * 256;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(i & 3), lane = (int)((i >> 2) & 63);
    const int64_t g = i >> 8;
    const int kg = (int)(g % kgroups), nt = (int)(g / kgroups);
    const int n = nt * 32 + (lane & 31), k = kg * 8 + (lane >> 5) * 4 + t;
    wp[i] = (n < d_out && k < d_in) ? w[(int64_t)n * ldw + k] : 0.f;
  }
}

template <int LPR, int U>
int launch_hub_gather(const SpmmArgs& a, int mode, int n_seg, hipStream_t st) {
  const bool cs = a.col_scale != nullptr;
  const dim3 g((unsigned)n_seg);
  if (mode == GLNN_AGG_SAGE_GCN && a.xf_on) hipLaunchKernelGGL((hub_gather_kernel<LPR, U, false, true>), g, dim3(kBlock), 0, st, a);
  else if (cs) hipLaunchKernelGGL((hub_gather_kernel<LPR, U, true, false>), g, dim3(kBlock), 0, st, a);
  else hipLaunchKernelGGL((hub_gather_kernel<LPR, U, false, false>), g, dim3(kBlock), 0, st, a);
  return glnn::check_launch("glnn_spmm_csr_f32(hub segments)");
}

template <int LPR, int U>
int launch_lpr(const SpmmArgs& a, int mode, hipStream_t st, int grid, int col_tiles = 1, int hub_segs = 0, bool short_rows = false) {
  const bool cs = a.col_scale != nullptr;
  const dim3 g(grid, col_tiles);
  if (hub_segs > 0) {                    // the hub rows' segments first: one workgroup each, into the plan's slab
    const int rc = launch_hub_gather<LPR, U>(a, mode, hub_segs, st);
    if (rc != GLNN_OK) return rc;
  }
  if constexpr (LPR >= 32) {             // (rows of more than 64 floats: the widths of the training blocks)
    // measured on the products training configuration (profiles/r05_spmm_short.txt): the plain SAGE aggregation of the outermost block (5
    // in-edges per row, 400-byte rows) 365 -> 294 us; the transposed blocks (1-2 in-edges, 1 KB rows written: bandwidth-bound already) equal;
    // the tail-in-gather launches (10-15 in-edges, most of a row's edges behind the batch) slower -- those two keep the row kernel
    if (short_rows && col_tiles == 1 && mode == GLNN_AGG_SAGE_GCN && !a.xf_on && a.cm.n == 0) {
      hipLaunchKernelGGL((spmm_csr_short_kernel<LPR, U, GLNN_AGG_SAGE_GCN, false>), g, dim3(kBlock), 0, st, a);
      return glnn::check_launch("glnn_spmm_csr_f32(short rows)");
    }
  }
  if (a.cm.n > 0) {                      // the chunks of a row range in ONE launch (glnn_spmm_csr_chunks_f32): the plain SAGE aggregation only
    if (mode != GLNN_AGG_SAGE_GCN || a.xf_on || col_tiles != 1)
      return glnn::fail(GLNN_ERR_UNSUPPORTED, "glnn_spmm_csr_chunks_f32: SAGE_GCN rows of <= 256 floats without a source transform only");
    hipLaunchKernelGGL((spmm_csr_kernel<LPR, U, GLNN_AGG_SAGE_GCN, false, false, true>), g, dim3(kBlock), 0, st, a);
    return glnn::check_launch("glnn_spmm_csr_chunks_f32");
  }
  if (mode == GLNN_AGG_SAGE_GCN && a.xf_on) {
    hipLaunchKernelGGL((spmm_csr_kernel<LPR, U, GLNN_AGG_SAGE_GCN, false, true>), g, dim3(kBlock), 0, st, a);
  } else if (mode == GLNN_AGG_SAGE_GCN) {
    hipLaunchKernelGGL((spmm_csr_kernel<LPR, U, GLNN_AGG_SAGE_GCN, false>), g, dim3(kBlock), 0, st, a);
  } else if (cs) {
    hipLaunchKernelGGL((spmm_csr_kernel<LPR, U, GLNN_AGG_SUM, true>), g, dim3(kBlock), 0, st, a);
  } else {
    hipLaunchKernelGGL((spmm_csr_kernel<LPR, U, GLNN_AGG_SUM, false>), g, dim3(kBlock), 0, st, a);
  }
  return glnn::check_launch("glnn_spmm_csr_f32");
}

__device__ __forceinline__ float degree_transform(float deg, int transform) {
  if (transform == GLNN_DEG_RSQRT_CLAMP1) return rsqrtf(fmaxf(deg, 1.0f));     // deg.clamp(min=1) ** -0.5
  if (transform == GLNN_DEG_INV_PLUS1) return 1.0f / (deg + 1.0f);
  return deg;
}

__global__ void degrees_kernel(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices, int64_t n_dst,
                               int64_t nnz, float* in_deg, float* out_deg_as_int, int transform) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (in_deg)
    for (int64_t v = i; v < n_dst; v += stride) in_deg[v] = degree_transform((float)(indptr[v + 1] - indptr[v]), transform);
  if (out_deg_as_int) {
    int* cnt = reinterpret_cast<int*>(out_deg_as_int);
    for (int64_t e = i; e < nnz; e += stride) atomicAdd(&cnt[indices[e]], 1);
  }
}

__global__ void int_to_float_kernel(float* p, int64_t n, int transform) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = degree_transform((float)reinterpret_cast<int*>(p)[i], transform);
}

}  // namespace

// a caller's glnn_hub_plan -> the kernels' HubPlan (d = the row width the slab must hold); *n_seg = segments to gather (0: no plan)
static int hub_plan_of(const glnn_hub_plan* plan, int d, HubPlan* h, int* n_seg, const char* who) {
  h->rows = nullptr; h->seg_ptr = nullptr; h->n_hub = 0; h->slab = nullptr; h->ld_slab = 0;
  *n_seg = 0;
  if (!plan || plan->n_hub == 0) return GLNN_OK;
  const int dpad = (d + 3) & ~3;
  GLNN_REQUIRE(plan->n_hub > 0 && plan->n_seg >= plan->n_hub && plan->rows && plan->seg_ptr && plan->slab, "%s: incomplete hub plan", who);
  GLNN_REQUIRE(plan->ld_slab % 4 == 0 && plan->ld_slab >= dpad && plan->slab_rows >= 8 * (int64_t)plan->n_seg && glnn::aligned16(plan->slab),
               "%s: the hub plan's slab needs >= %d rows (8 per segment) of >= %d floats (ld multiple of 4), 16-byte aligned", who, 8 * plan->n_seg, dpad);
  h->rows = plan->rows; h->seg_ptr = plan->seg_ptr; h->n_hub = plan->n_hub; h->slab = plan->slab; h->ld_slab = plan->ld_slab;
  *n_seg = plan->n_seg;
  return GLNN_OK;
}

// glnn_chunk_signals -> the kernels' ChunkMap (tile = 32 rows); chunks == NULL: off
static int fill_chunk_map(const glnn_chunk_signals* chunks, int64_t n_dst, ChunkMap* cm, const char* who) {
  *cm = ChunkMap{};
  if (!chunks) return GLNN_OK;
  const int nc = chunks->n_chunks;
  GLNN_REQUIRE(nc >= 1 && nc <= GLNN_MAX_CHUNKS && chunks->arrivals && chunks->row_start[0] == 0 && chunks->row_start[nc] >= n_dst,
               "%s: 1..%d chunks covering rows [0, n_dst), arrival counters", who, GLNN_MAX_CHUNKS);
  const int64_t tiles = (n_dst + 31) / 32;
  GLNN_REQUIRE(tiles < ((int64_t)1 << 31), "%s: n_dst too large for one launch", who);
  cm->n = nc; cm->arrivals = chunks->arrivals; cm->epoch = chunks->epoch;
  for (int c = 0; c < nc; ++c) {
    const int64_t r0 = chunks->row_start[c], r1 = chunks->row_start[c + 1];
    GLNN_REQUIRE(r0 % 32 == 0 && r1 >= r0 && chunks->signal[c] && chunks->self_row[c] >= 0 && chunks->out_row[c] >= 0,
                 "%s: chunk %d: row_start must be an ascending multiple of 32, signal / rows set", who, c);
    cm->tile_start[c] = (int)(r0 / 32 < tiles ? r0 / 32 : tiles);
    cm->self_shift[c] = chunks->self_row[c] - r0;
    cm->out_shift[c] = chunks->out_row[c] - r0;
    cm->signal[c] = chunks->signal[c];
  }
  for (int c = nc; c <= GLNN_MAX_CHUNKS; ++c) cm->tile_start[c] = (int)tiles;
  return GLNN_OK;
}

static int spmm_impl(const int64_t* indptr, const int32_t* indices, int64_t n_dst, int64_t n_src,
                     const float* x, int64_t ldx, int d, int mode, const float* row_scale,
                     const float* col_scale, const float* x_self, int64_t ld_self, const int64_t* self_rows,
                     const float* ep_scale, const float* ep_shift, int relu, float* out, int64_t ldo,
                     void* stream, const glnn::SourceTail* tail, const glnn_hub_plan* plan = nullptr, int64_t nnz_hint = -1,
                     const glnn_chunk_signals* chunks = nullptr) {
  if (n_dst == 0) return GLNN_OK;                       // nothing to do (empty tensors carry null pointers)
  GLNN_REQUIRE(indptr && x && out, "glnn_spmm_csr_f32: null pointer");   // indices may be NULL iff the graph has no edges
  GLNN_REQUIRE(n_dst >= 0 && n_src >= 0 && n_src < (int64_t)1 << 31, "glnn_spmm_csr_f32: bad n_dst/n_src");
  GLNN_REQUIRE(d >= 1, "glnn_spmm_csr_f32: d=%d must be >= 1", d);
  GLNN_REQUIRE(mode == GLNN_AGG_SUM || mode == GLNN_AGG_SAGE_GCN, "glnn_spmm_csr_f32: unknown mode %d", mode);
  const int dpad = (d + 3) & ~3;
  GLNN_REQUIRE(ldx % 4 == 0 && ldx >= dpad, "glnn_spmm_csr_f32: ldx=%lld must be a multiple of 4 and >= %d", (long long)ldx, dpad);
  GLNN_REQUIRE(ldo % 4 == 0 && ldo >= dpad, "glnn_spmm_csr_f32: ldo=%lld must be a multiple of 4 and >= %d", (long long)ldo, dpad);
  GLNN_REQUIRE(glnn::aligned16(x) && glnn::aligned16(out), "glnn_spmm_csr_f32: x/out must be 16-byte aligned");
  if (mode == GLNN_AGG_SAGE_GCN) {
    GLNN_REQUIRE(x_self && ld_self % 4 == 0 && ld_self >= dpad && glnn::aligned16(x_self),
                 "glnn_spmm_csr_f32: SAGE_GCN needs x_self with ld multiple of 4");
    GLNN_REQUIRE(!row_scale && !col_scale, "glnn_spmm_csr_f32: scales are not used in SAGE_GCN mode");
  } else {
    GLNN_REQUIRE(!self_rows, "glnn_spmm_csr_f32: self_rows belongs to SAGE_GCN mode");
  }
  if (n_dst == 0) return GLNN_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);

  // column tiles of <= 256 floats (64 lanes x float4); wider rows (raw cora / citeseer features) take ONE launch whose blockIdx.y is the
  // tile (the kernel shifts its pointers): `wide` runs this loop body once with the whole width
  const bool wide = d > 256 && (d + 255) / 256 <= 65535;
  if (tail) {
    GLNN_REQUIRE(mode == GLNN_AGG_SAGE_GCN && d <= 256, "glnn::spmm_csr_tail: SAGE_GCN rows of <= 256 floats only");
    GLNN_REQUIRE((tail->scale == nullptr) == (tail->shift == nullptr) && tail->drop_p >= 0.f && tail->drop_p < 1.f, "glnn::spmm_csr_tail: bad tail");
  }
  for (int c0 = 0; c0 < d; c0 += (wide ? d : 256)) {
    const int dt = wide ? 256 : ((d - c0) < 256 ? (d - c0) : 256);
    SpmmArgs a;
    a.indptr = indptr; a.indices = indices; a.n_dst = n_dst;
    a.x = x + c0; a.ldx = ldx; a.d = wide ? d : dt;
    a.row_scale = row_scale; a.col_scale = col_scale;
    a.x_self = x_self ? x_self + c0 : nullptr; a.ld_self = ld_self; a.self_rows = self_rows;
    a.ep_scale = ep_scale ? ep_scale + c0 : nullptr; a.ep_shift = ep_shift ? ep_shift + c0 : nullptr;
    a.relu = relu; a.out = out + c0; a.ldo = ldo;
    a.xf_on = tail ? 1 : 0;
    a.xf_scale = tail ? tail->scale : nullptr; a.xf_shift = tail ? tail->shift : nullptr;
    a.xf_thr = tail ? glnn::drop_threshold(tail->drop_p) : 0u; a.xf_seed = tail ? tail->drop_seed : 0u;
    a.xf_dscale = tail ? 1.0f / (1.0f - tail->drop_p) : 1.f;
    {
      GLNN_REQUIRE(!chunks || (d <= 256 && mode == GLNN_AGG_SAGE_GCN && !tail), "glnn_spmm_csr_chunks_f32: SAGE_GCN rows of <= 256 floats only");
      const int rcm = fill_chunk_map(chunks, n_dst, &a.cm, "glnn_spmm_csr_chunks_f32");
      if (rcm != GLNN_OK) return rcm;
    }
    int hub_segs = 0;
    {
      const int rch = hub_plan_of(d <= 256 ? plan : nullptr, d, &a.hub, &hub_segs, "glnn_spmm_csr_f32");      // (column-tiled rows: no plan)
      if (rch != GLNN_OK) return rch;
    }
    // workgroups in the long-row role: one per 512-row scan chunk, at most GLNN_LONG_BLOCK_CAP.  (Until round 4: n_dst / 4096 -- right
    // for a whole graph, where it hits the cap, but a row SHARD of a power-law graph keeps the graph's hub rows: rank r of 8 launches
    // 76 k-row chunks of the products graph whose rows of degree > 128 hold 20 % of the edges, and 18 workgroups gathered them while
    // the other 2,400 had long finished -- 5.1 ms for an eighth of the 8.8 ms whole-graph launch, profiles/scale_model_r04_a.json)
    int64_t n_long = (n_dst + GLNN_LONG_BLOCK_ROWS - 1) / GLNN_LONG_BLOCK_ROWS;
    if (n_long < 1) n_long = 1;
    if (n_long > GLNN_LONG_BLOCK_CAP) n_long = GLNN_LONG_BLOCK_CAP;
    a.n_long_blocks = (int)n_long;
    // rows per wave: 16 on whole graphs (amortises the workgroup's set-up; chosen by sweep), fewer on small inputs -- the
    // blocks of a training batch have 0.5k-50k rows, and 16 dependent rows per wave left most of the 1024 SIMDs idle
    // (80 us for a 7k-row block): keep >= ~2048 workgroups in flight
    int64_t rpw = n_dst / (2048 * kWavesPerBlock);
    if (rpw < 1) rpw = 1;
    if (rpw > kRowsPerWave) rpw = kRowsPerWave;
    if (a.cm.n > 0) {
      // no row workgroup may straddle two chunks: rows per workgroup = the largest power of two <= the usual count that divides every
      // chunk's first row (chunks start at multiples of 32: 4 rows per wave always does)
      int64_t p = 1;
      while (p * 2 <= rpw) p *= 2;
      for (int c = 1; c < a.cm.n; ++c)
        while (p > 1 && ((int64_t)a.cm.tile_start[c] * 32) % (kWavesPerBlock * p) != 0) p /= 2;
      rpw = p;
    }
    const int64_t rows_per_block = kWavesPerBlock * rpw;
    a.rows_per_block = (int)rows_per_block;
    const int64_t row_blocks = (n_dst + rows_per_block - 1) / rows_per_block;
    GLNN_REQUIRE(row_blocks + n_long < ((int64_t)1 << 31), "glnn_spmm_csr_f32: n_dst too large for one launch");
    const int grid = (int)(row_blocks + n_long);
    const int dv = (dt + 3) / 4;
    // a caller that knows the edge count of a sparse block (<= 6 in-edges per row on average: a row's whole gather fits the batch) gets
    // the short-row kernel -- the same bits, four rows per wave in flight (GLNN_SPMM_SHORT=0: never)
    const bool short_rows = nnz_hint >= 0 && nnz_hint <= 6 * n_dst && glnn::opts().spmm_short != 0 && a.rows_per_block % kShortRows == 0;
    int rc;
    if (dv <= 4) rc = launch_lpr<4, GLNN_SPMM_U>(a, mode, st, grid, 1, hub_segs);
    else if (dv <= 8) rc = launch_lpr<8, GLNN_SPMM_U>(a, mode, st, grid, 1, hub_segs);
    else if (dv <= 16) rc = launch_lpr<16, GLNN_SPMM_U>(a, mode, st, grid, 1, hub_segs);
    else if (dv <= 32) rc = launch_lpr<32, GLNN_SPMM_U>(a, mode, st, grid, 1, hub_segs, short_rows);
    else rc = launch_lpr<64, GLNN_SPMM_U>(a, mode, st, grid, wide ? (d + 255) / 256 : 1, hub_segs, short_rows);
    if (rc != GLNN_OK) return rc;
  }
  return GLNN_OK;
}

extern "C" int glnn_spmm_csr_f32(const int64_t* indptr, const int32_t* indices, int64_t n_dst, int64_t n_src,
                                 const float* x, int64_t ldx, int d, int mode, const float* row_scale,
                                 const float* col_scale, const float* x_self, int64_t ld_self, const int64_t* self_rows,
                                 const float* ep_scale, const float* ep_shift, int relu, float* out, int64_t ldo,
                                 void* stream) {
  return spmm_impl(indptr, indices, n_dst, n_src, x, ldx, d, mode, row_scale, col_scale, x_self, ld_self, self_rows, ep_scale, ep_shift, relu,
                   out, ldo, stream, nullptr);
}

extern "C" int glnn_spmm_csr_plan_f32(const int64_t* indptr, const int32_t* indices, int64_t n_dst, int64_t n_src,
                                      const float* x, int64_t ldx, int d, int mode, const float* row_scale,
                                      const float* col_scale, const float* x_self, int64_t ld_self, const int64_t* self_rows,
                                      const float* ep_scale, const float* ep_shift, int relu, float* out, int64_t ldo,
                                      const glnn_hub_plan* plan, void* stream) {
  return spmm_impl(indptr, indices, n_dst, n_src, x, ldx, d, mode, row_scale, col_scale, x_self, ld_self, self_rows, ep_scale, ep_shift, relu,
                   out, ldo, stream, nullptr, plan);
}

extern "C" int glnn_spmm_csr_chunks_f32(const int64_t* indptr, const int32_t* indices, int64_t n_dst, int64_t n_src,
                                        const float* x, int64_t ldx, int d, int mode, const float* row_scale,
                                        const float* col_scale, const float* x_self, int64_t ld_self, const int64_t* self_rows,
                                        const float* ep_scale, const float* ep_shift, int relu, float* out, int64_t ldo,
                                        const glnn_hub_plan* plan, const glnn_chunk_signals* chunks, void* stream) {
  GLNN_REQUIRE(chunks, "glnn_spmm_csr_chunks_f32: chunks is NULL (use glnn_spmm_csr_plan_f32)");
  return spmm_impl(indptr, indices, n_dst, n_src, x, ldx, d, mode, row_scale, col_scale, x_self, ld_self, self_rows, ep_scale, ep_shift, relu,
                   out, ldo, stream, nullptr, plan, -1, chunks);
}

extern "C" int glnn_hub_row_threshold(void) { return kHubRow; }
extern "C" int glnn_hub_segment_edges(void) { return kHubSeg; }

// SAGE-"gcn" aggregation of rows that exist only as pre-activations z: every gathered / self row is tail(z) = drop(relu(z * scale +
// shift)) evaluated in the gather (glnn_sage_fwd_bwd_f32: the hidden layers' h = tail(z) is never written)
int glnn::spmm_csr_tail(const int64_t* indptr, const int32_t* indices, int64_t n_dst, int64_t n_src, const float* z, int64_t ldz, int d,
                        const glnn::SourceTail& tail, float* out, int64_t ldo, void* stream, int64_t nnz) {
  return spmm_impl(indptr, indices, n_dst, n_src, z, ldz, d, GLNN_AGG_SAGE_GCN, nullptr, nullptr, z, ldz, nullptr, nullptr, nullptr, 0, out, ldo,
                   stream, &tail, nullptr, nnz);
}

// glnn_spmm_csr_f32 for a caller that knows the block's edge count (glnn_sage_fwd_bwd_f32): sparse blocks take the short-row kernel
int glnn::spmm_csr_nnz(const int64_t* indptr, const int32_t* indices, int64_t n_dst, int64_t n_src, int64_t nnz, const float* x, int64_t ldx, int d,
                       int mode, const float* col_scale, const float* x_self, int64_t ld_self, const int64_t* self_rows, float* out, int64_t ldo,
                       void* stream) {
  return spmm_impl(indptr, indices, n_dst, n_src, x, ldx, d, mode, nullptr, col_scale, x_self, ld_self, self_rows, nullptr, nullptr, 0, out, ldo,
                   stream, nullptr, nullptr, nnz);
}

// the launch shape of spmm_bn_dy_kernel over n_dst rows: long-row workgroups + row workgroups = the number of column-sum slots
static int64_t bn_dy_grid(int64_t n_dst, int* n_long_blocks, int* rows_per_block) {
  int64_t n_long = (n_dst + GLNN_LONG_BLOCK_ROWS - 1) / GLNN_LONG_BLOCK_ROWS;
  if (n_long < 1) n_long = 1;
  if (n_long > GLNN_LONG_BLOCK_CAP) n_long = GLNN_LONG_BLOCK_CAP;
  int64_t rpw = n_dst / (2048 * kWavesPerBlock);
  if (rpw < 4) rpw = 4;                                   // (>= 32 rows per workgroup: its 2 x d floats of column sums stay a few % of what it writes)
  if (rpw > kRowsPerWave) rpw = kRowsPerWave;
  const int64_t rpb = kWavesPerBlock * rpw;
  if (n_long_blocks) *n_long_blocks = (int)n_long;
  if (rows_per_block) *rows_per_block = (int)rpb;
  return (n_dst + rpb - 1) / rpb + n_long;
}
extern "C" int64_t glnn_sage_step_ws_bn_floats(int64_t n_dst_0, int hidden) {
  if (n_dst_0 < 1 || hidden < 1) return 0;
  const int64_t h = (hidden + 3) & ~3;
  return 2 * bn_dy_grid(n_dst_0, nullptr, nullptr) * h + 5 * h + 8;
}

// out = dy of (A x col_scale) behind the BatchNorm tail described by `tail`, plus the per-workgroup column sums for that BatchNorm's
// backward: see spmm_bn_dy_kernel.  ws: 2 * (*nslots) * d floats (ws1 = ws, ws2 = ws + nslots d).  GLNN_ERR_UNSUPPORTED with nothing
// launched unless 64 < d <= 256, d % 4 == 0, float4-addressable rows, and ws holds the slots.
int glnn::spmm_csr_bn_dy(const int64_t* indptr, const int32_t* indices, int64_t n_dst, int64_t n_src, const float* x, int64_t ldx, int d,
                         const float* col_scale, const glnn::BnTail& tail, float* out, int64_t ldo, float* ws, int64_t ws_floats, int* nslots,
                         void* stream) {
  if (!indptr || !x || !out || !col_scale || !ws || !nslots || n_dst < 1 || n_src < 0 || n_src >= ((int64_t)1 << 31)) return GLNN_ERR_UNSUPPORTED;
  if (d <= 64 || d > 256 || (d & 3) || (ldx & 3) || (ldo & 3) || (tail.ldz & 3) || ldx < d || ldo < d || tail.ldz < d) return GLNN_ERR_UNSUPPORTED;
  if (!tail.z || !tail.mean || !tail.rstd || !tail.a_scale || !tail.a_shift || tail.drop_p < 0.f || tail.drop_p >= 1.f) return GLNN_ERR_UNSUPPORTED;
  if (!glnn::aligned16(x) || !glnn::aligned16(out) || !glnn::aligned16(tail.z) || !glnn::aligned16(tail.mean) || !glnn::aligned16(tail.rstd) ||
      !glnn::aligned16(tail.a_scale) || !glnn::aligned16(tail.a_shift) || !glnn::aligned16(ws))
    return GLNN_ERR_UNSUPPORTED;
  SpmmArgs a = {};
  a.indptr = indptr; a.indices = indices; a.n_dst = n_dst; a.x = x; a.ldx = ldx; a.d = d; a.col_scale = col_scale; a.out = out; a.ldo = ldo;
  const int64_t slots = bn_dy_grid(n_dst, &a.n_long_blocks, &a.rows_per_block);
  if (slots >= ((int64_t)1 << 24) || 2 * slots * d > ws_floats) return GLNN_ERR_UNSUPPORTED;
  DyTail t;
  t.z = tail.z; t.ldz = tail.ldz; t.mean = tail.mean; t.rstd = tail.rstd; t.a_scale = tail.a_scale; t.a_shift = tail.a_shift;
  t.dthr = glnn::drop_threshold(tail.drop_p); t.dseed = tail.drop_seed; t.dscale = 1.0f / (1.0f - tail.drop_p); t.relu = tail.relu;
  t.ws1 = ws; t.ws2 = ws + slots * d; t.h = d;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 g((unsigned)slots);
  if (d > 128) {
    if (t.dthr) hipLaunchKernelGGL((spmm_bn_dy_kernel<64, GLNN_SPMM_U, true>), g, dim3(kBlock), 0, st, a, t);
    else hipLaunchKernelGGL((spmm_bn_dy_kernel<64, GLNN_SPMM_U, false>), g, dim3(kBlock), 0, st, a, t);
  } else {
    if (t.dthr) hipLaunchKernelGGL((spmm_bn_dy_kernel<32, GLNN_SPMM_U, true>), g, dim3(kBlock), 0, st, a, t);
    else hipLaunchKernelGGL((spmm_bn_dy_kernel<32, GLNN_SPMM_U, false>), g, dim3(kBlock), 0, st, a, t);
  }
  *nslots = (int)slots;
  return glnn::check_launch("glnn::spmm_csr_bn_dy");
}

extern "C" int glnn_degrees_f32(const int64_t* indptr, const int32_t* indices, int64_t n_dst, int64_t n_src,
                                int64_t nnz, int transform, float* in_deg, float* out_deg, void* stream) {
  GLNN_REQUIRE(indptr, "glnn_degrees_f32: null indptr");
  GLNN_REQUIRE(transform >= GLNN_DEG_RAW && transform <= GLNN_DEG_INV_PLUS1, "glnn_degrees_f32: unknown transform %d", transform);
  GLNN_REQUIRE(!out_deg || indices, "glnn_degrees_f32: out_deg needs indices");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  GLNN_REQUIRE(nnz >= 0 && n_dst >= 0 && n_src >= 0, "glnn_degrees_f32: negative size");
  if (out_deg) {
    if (hipMemsetAsync(out_deg, 0, sizeof(float) * (size_t)n_src, st) != hipSuccess)
      return glnn::fail(GLNN_ERR_HIP, "glnn_degrees_f32: memset failed");
  }
  const int64_t work = nnz > n_dst ? nnz : n_dst;
  int grid = (int)((work + 255) / 256);
  if (grid > 4096) grid = 4096;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(degrees_kernel, dim3(grid), dim3(256), 0, st, indptr, indices, n_dst, nnz, in_deg, out_deg, transform);
  if (out_deg && n_src > 0)
    hipLaunchKernelGGL(int_to_float_kernel, dim3((unsigned)((n_src + 255) / 256)), dim3(256), 0, st, out_deg, n_src, transform);
  return glnn::check_launch("glnn_degrees_f32");
}

extern "C" int64_t glnn_packed_weight_floats(int d_out, int d_in) {
  return (int64_t)((d_out + 31) / 32) * ((d_in + 7) / 8) * 256;
}

extern "C" int glnn_pack_weight_f32(const float* w, int64_t ldw, int d_out, int d_in, float* w_packed, void* stream) {
  GLNN_REQUIRE(w && w_packed && d_out >= 1 && d_in >= 1 && ldw >= d_in, "glnn_pack_weight_f32: bad arguments");
  GLNN_REQUIRE(glnn::aligned16(w_packed), "glnn_pack_weight_f32: w_packed must be 16-byte aligned");
  const int kgroups = (d_in + 7) / 8;
  const int64_t total = glnn_packed_weight_floats(d_out, d_in);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, ldw, d_out,
                     d_in, kgroups, w_packed);
  return glnn::check_launch("glnn_pack_weight_f32");
}

static int sage_fused_impl(const int64_t* indptr, const int32_t* indices, int64_t n_dst, int64_t n_src, const float* x,
                           int64_t ldx, int d_in, const float* x_self, int64_t ld_self, const float* w_packed,
                           int d_out, const float* ep_scale, const float* ep_shift, int relu, float* out,
                           int64_t ldo, const float* w2_packed, int d_out2, float* out2, int64_t ldo2, const int32_t* tile_order,
                           const glnn_hub_plan* plan, void* stream, const glnn_chunk_signals* chunks = nullptr) {
  if (n_dst == 0) return GLNN_OK;
  GLNN_REQUIRE(indptr && x && x_self && w_packed && (out || w2_packed), "glnn_sage_fused_f32: null pointer");   // indices NULL iff no edges
  GLNN_REQUIRE(!w2_packed || (out2 && d_out2 >= 1 && d_out2 <= 256 && ldo2 >= d_out2 && glnn::aligned16(w2_packed)),
               "glnn_sage_fused_f32: the chained projection needs out2 with ldo2 >= d_out2 in [1,256]");
  GLNN_REQUIRE(n_dst >= 0 && n_src >= 0 && n_src < (int64_t)1 << 31, "glnn_sage_fused_f32: bad n_dst/n_src");
  GLNN_REQUIRE(d_in >= 1 && d_in <= 256 && d_out >= 1 && d_out <= 256, "glnn_sage_fused_f32: d_in and d_out must be in [1,256]");
  const int dpad = (d_in + 3) & ~3;
  GLNN_REQUIRE(ldx % 4 == 0 && ldx >= dpad && ld_self % 4 == 0 && ld_self >= dpad && (!out || ldo >= d_out),
               "glnn_sage_fused_f32: leading dimensions (ldx, ld_self multiples of 4 and >= %d; ldo >= d_out)", dpad);
  GLNN_REQUIRE(glnn::aligned16(x) && glnn::aligned16(x_self) && glnn::aligned16(w_packed), "glnn_sage_fused_f32: 16-byte alignment required");
  if (n_dst == 0) return GLNN_OK;
  FusedArgs a;
  a.indptr = indptr; a.indices = indices; a.n_dst = n_dst; a.x = x; a.ldx = ldx; a.d_in = d_in; a.x_self = x_self;
  a.ld_self = ld_self; a.w_packed = w_packed; a.d_out = d_out; a.kgroups = (d_in + 7) / 8; a.ep_scale = ep_scale;
  a.ep_shift = ep_shift; a.relu = relu; a.out = out; a.ldo = ldo;
  a.w2_packed = w2_packed; a.d_out2 = w2_packed ? d_out2 : 0; a.kgroups2 = w2_packed ? (d_out + 7) / 8 : 0; a.out2 = out2; a.ldo2 = ldo2;
  a.tile_order = tile_order;
  {
    const int rcm = fill_chunk_map(chunks, n_dst, &a.cm, "glnn_sage_fused_chunks_f32");
    if (rcm != GLNN_OK) return rcm;
  }
  // one 32-row sub-tile per workgroup (RT = 1).  RT = 2 (64-row tiles, every W fragment load feeding two MFMA chains) was measured
  // 1.5x slower in round 1 -- big workgroups drain badly -- and its instantiations were removed in round 3
  constexpr int rt = 1;
  const int rows_per_wg = 32 * rt;
  const int64_t blocks = (n_dst + rows_per_wg - 1) / rows_per_wg;
  GLNN_REQUIRE(blocks < ((int64_t)1 << 31), "glnn_sage_fused_f32: n_dst too large for one launch");
  const int kg_lds = a.kgroups2 > a.kgroups ? a.kgroups2 : a.kgroups;      // the hidden tile of the chained pass reuses the aggregate tile
  const size_t smem = sizeof(float) * rows_per_wg * (kg_lds * 8 + 4);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int dv = dpad / 4;
  // columns [4*LPR, kpad) must not exist: LPR*4 >= kpad is guaranteed by picking LPR from kpad (a multiple of 8)
  const int kv = a.kgroups * 2;      // float4 per padded row
  int hub_segs = 0;
  {
    const int rch = hub_plan_of(plan, d_in, &a.hub, &hub_segs, "glnn_sage_fused_f32");
    if (rch != GLNN_OK) return rch;
  }
  if (hub_segs > 0) {
    // the hub rows' segments first (one workgroup each, the lane mapping of the fused launch: same sums as its own in-tile path)
    SpmmArgs h = {};
    h.indptr = indptr; h.indices = indices; h.n_dst = n_dst; h.x = x; h.ldx = ldx; h.d = d_in; h.hub = a.hub;
    int rc;
    if (kv <= 16 && dv <= 16) rc = launch_hub_gather<16, GLNN_FUSED_U>(h, GLNN_AGG_SUM, hub_segs, st);
    else if (kv <= 32) rc = launch_hub_gather<32, GLNN_FUSED_U>(h, GLNN_AGG_SUM, hub_segs, st);
    else rc = launch_hub_gather<64, GLNN_FUSED_U>(h, GLNN_AGG_SUM, hub_segs, st);
    if (rc != GLNN_OK) return rc;
  }
#define GLNN_FUSED_LAUNCH(LPR_, RT_)                                                                                                          \
  do {                                                                                                                                        \
    if (a.cm.n) hipLaunchKernelGGL((sage_fused_kernel<LPR_, GLNN_FUSED_U, RT_, true>), dim3((unsigned)blocks), dim3(kFusedBlock), smem, st, a); \
    else hipLaunchKernelGGL((sage_fused_kernel<LPR_, GLNN_FUSED_U, RT_>), dim3((unsigned)blocks), dim3(kFusedBlock), smem, st, a);             \
  } while (0)
  if (kv <= 16 && dv <= 16) GLNN_FUSED_LAUNCH(16, 1); else if (kv <= 32) GLNN_FUSED_LAUNCH(32, 1); else GLNN_FUSED_LAUNCH(64, 1);
#undef GLNN_FUSED_LAUNCH
  return glnn::check_launch("glnn_sage_fused_f32");
}

extern "C" int glnn_sage_fused_f32(const int64_t* indptr, const int32_t* indices, int64_t n_dst, int64_t n_src, const float* x,
                                   int64_t ldx, int d_in, const float* x_self, int64_t ld_self, const float* w_packed,
                                   int d_out, const float* ep_scale, const float* ep_shift, int relu, float* out,
                                   int64_t ldo, const float* w2_packed, int d_out2, float* out2, int64_t ldo2, const int32_t* tile_order,
                                   void* stream) {
  return sage_fused_impl(indptr, indices, n_dst, n_src, x, ldx, d_in, x_self, ld_self, w_packed, d_out, ep_scale, ep_shift, relu, out, ldo,
                         w2_packed, d_out2, out2, ldo2, tile_order, nullptr, stream);
}

extern "C" int glnn_sage_fused_plan_f32(const int64_t* indptr, const int32_t* indices, int64_t n_dst, int64_t n_src, const float* x,
                                        int64_t ldx, int d_in, const float* x_self, int64_t ld_self, const float* w_packed,
                                        int d_out, const float* ep_scale, const float* ep_shift, int relu, float* out,
                                        int64_t ldo, const float* w2_packed, int d_out2, float* out2, int64_t ldo2, const int32_t* tile_order,
                                        const glnn_hub_plan* plan, void* stream) {
  return sage_fused_impl(indptr, indices, n_dst, n_src, x, ldx, d_in, x_self, ld_self, w_packed, d_out, ep_scale, ep_shift, relu, out, ldo,
                         w2_packed, d_out2, out2, ldo2, tile_order, plan, stream);
}

extern "C" int glnn_sage_fused_chunks_f32(const int64_t* indptr, const int32_t* indices, int64_t n_dst, int64_t n_src, const float* x,
                                          int64_t ldx, int d_in, const float* x_self, int64_t ld_self, const float* w_packed,
                                          int d_out, const float* ep_scale, const float* ep_shift, int relu, float* out,
                                          int64_t ldo, const float* w2_packed, int d_out2, float* out2, int64_t ldo2, const int32_t* tile_order,
                                          const glnn_hub_plan* plan, const glnn_chunk_signals* chunks, void* stream) {
  GLNN_REQUIRE(chunks, "glnn_sage_fused_chunks_f32: chunks is NULL (use glnn_sage_fused_plan_f32)");
  return sage_fused_impl(indptr, indices, n_dst, n_src, x, ldx, d_in, x_self, ld_self, w_packed, d_out, ep_scale, ep_shift, relu, out, ldo,
                         w2_packed, d_out2, out2, ldo2, tile_order, plan, stream, chunks);
}

extern "C" int glnn_signal_alloc(uint32_t** signal) {
  GLNN_REQUIRE(signal, "glnn_signal_alloc: null pointer");
  void* p = nullptr;
  if (hipExtMallocWithFlags(&p, 8, hipMallocSignalMemory) != hipSuccess || !p) {
    (void)hipGetLastError();
    return glnn::fail(GLNN_ERR_HIP, "glnn_signal_alloc: hipExtMallocWithFlags(hipMallocSignalMemory) failed");
  }
  if (hipMemset(p, 0, 8) != hipSuccess) return glnn::fail(GLNN_ERR_HIP, "glnn_signal_alloc: memset failed");
  *signal = reinterpret_cast<uint32_t*>(p);
  return GLNN_OK;
}

extern "C" int glnn_signal_free(uint32_t* signal) {
  if (signal && hipFree(signal) != hipSuccess) return glnn::fail(GLNN_ERR_HIP, "glnn_signal_free: hipFree failed");
  return GLNN_OK;
}

extern "C" int glnn_signal_read(const uint32_t* signal, uint32_t* value) {
  GLNN_REQUIRE(signal && value, "glnn_signal_read: null pointer");
  if (hipMemcpy(value, signal, 4, hipMemcpyDeviceToHost) != hipSuccess) return glnn::fail(GLNN_ERR_HIP, "glnn_signal_read: copy failed");
  return GLNN_OK;
}

__global__ void signal_fence_kernel() {}

extern "C" int glnn_stream_wait_value32(void* stream, uint32_t* signal, uint32_t value) {
  GLNN_REQUIRE(signal, "glnn_stream_wait_value32: null pointer");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (hipStreamWaitValue32(st, signal, value, hipStreamWaitValueGte, 0xffffffffu) != hipSuccess) {
    (void)hipGetLastError();
    return glnn::fail(GLNN_ERR_HIP, "glnn_stream_wait_value32: hipStreamWaitValue32 failed (not supported on this device?)");
  }
  // the signalled rows sit in the producing XCDs' L2s (see store_out): an empty kernel, whose end-of-kernel release writes the L2s back
  if (glnn::opts().signal_fence) {                                             // (GLNN_SIGNAL_NO_FENCE=1: the negative control of the tests)
    hipLaunchKernelGGL(signal_fence_kernel, dim3(1), dim3(64), 0, st);
    return glnn::check_launch("glnn_stream_wait_value32(fence)");
  }
  return GLNN_OK;
}
