// K3: fp32 dense projections on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: f32 in, f32
// accumulate, bit-equal to an fmaf chain, 157 TF peak -- the 1e-4 parity bar rules out bf16).
//
// Replaces nn.Linear / dgl SAGEConv.fc_neigh / dgl GraphConv.weight matmuls of the reference
// (models.py:45,112,138,193) and, in the TN form, the weight-gradient matmul autograd runs inside
// loss.backward() (train_and_eval.py:84).
//
// Tiling: 128 x BN block tile (BN = 128 or 64), BK = 32, 256 threads = 4 waves as 2(M) x 2(N); each
// wave owns 64 x BN/2 = 2 x (BN/64) MFMA 32x32 tiles (64 / 32 accumulator VGPRs).  Operand tiles go
// global -> registers -> LDS (double-buffered, one barrier per k-tile, next tile's global loads issued
// before the MFMAs of the current one).  The operand transform of the previous layer tail (row gather,
// BatchNorm scale/shift, ReLU) is applied in registers on the way to LDS, so activations are never
// re-materialised in HBM.
//
// k-order trick for the row-major-along-k operands: a lane reads 4 consecutive k with ONE ds_read_b128
// (lane half kk takes k0+4kk..k0+4kk+3) and feeds MFMA t with element t, i.e. the 4 MFMAs of a group
// consume k pairs (t, 4+t).  Both operands use the same pairing, and a dot product does not care about
// the order of its terms.  LDS rows are padded to 36 floats: conflict-free for ds_read_b128's 16-lane
// groups (slot = 9*i mod 16 is a bijection on each group).
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "glnn_common.h"

namespace {

// compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N-1>)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDS_K = BK + 4;  // padded row stride (floats) for [row][k] tiles

struct GemmArgs {
  const float* a; int64_t lda; const int64_t* a_rows; const float* a_scale; const float* a_shift;
  int64_t m; int k;
  const float* b; int64_t ldb; int n;
  const float* row_scale; const float* ep_scale; const float* ep_shift; int relu;
  float* c; int64_t ldc;
  int a_vec; int b_vec;  // 1 = float4 loads are aligned and in-bounds
  uint32_t drop_thr; uint32_t drop_seed; float drop_scale;   // drop_thr == 0: no dropout
  int ksplits; int ktiles_per_split; float* ws;               // split-K (fast kernel): raw partials -> ws[z][m][n]
  int defer_fold;     // host side only: the split-K partials are folded (sum over z + epilogue) by the CONSUMER of C, not by a launch here
  // gemm_kernel_pipe only (else NULL): per 128-row tile and column, mean / M2 of the stored values at st_*[tile_row * n + col] -- the first
  // pass of the BatchNorm statistics over C taken from the accumulators (glnn::gemm_stats)
  float* st_mean; float* st_m2;
  int* st_done;       // host side only: set to 1 by the launcher when the kernel that writes them was chosen
  // gemm_kernel_pipe<true> only (else bd_z == NULL; glnn::gemm_bn_dy): C is the input gradient da of a hidden layer whose tail is
  // BatchNorm -> ReLU -> dropout over the pre-activations bd_z.  The epilogue stores dy = da behind the tail's masks instead of da and
  // leaves, per 128-row tile and column, S1 = sum dy and S2 = sum dy * xhat at bd_s1 / bd_s2[tile_row * n + col]: the first pass of the
  // BatchNorm backward taken from the accumulators (it used to read da and z back: 64 MB for MLP3w8's first hidden layer)
  const float* bd_z; int64_t bd_ldz; const float* bd_scale; const float* bd_shift; const float* bd_mean; const float* bd_rstd;
  uint32_t bd_thr; uint32_t bd_seed; float bd_dscale; int bd_relu; float* bd_s1; float* bd_s2;
};

__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// 4 consecutive elements p[0..3] of a row, elements at index >= limit (relative to p) read as 0.
__device__ __forceinline__ float4 load4_guard(const float* p, int limit, bool vec) {
  if (limit <= 0) return zero4();
  if (vec && limit >= 4) return *reinterpret_cast<const float4*>(p);
  float4 r;
  r.x = p[0];
  r.y = limit > 1 ? p[1] : 0.f;
  r.z = limit > 2 ? p[2] : 0.f;
  r.w = limit > 3 ? p[3] : 0.f;
  return r;
}

struct DropCfg { uint32_t thr, seed; float scale; };

__device__ __forceinline__ float4 xform4(float4 v, const float* sc, const float* sh, int k, int klimit, int64_t row,
                                         const DropCfg dc) {
  // a' = drop(max(a*scale + shift, 0)) for k < klimit, else 0
  float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (k + t < klimit) {
      float y = fmaxf(fmaf(vv[t], sc[k + t], sh[k + t]), 0.f);
      if (dc.thr) y = glnn::drop_keep(dc.seed, dc.thr, (uint32_t)row, (uint32_t)(k + t)) ? y * dc.scale : 0.f;
      vv[t] = y;
    } else {
      vv[t] = 0.f;
    }
  }
  return make_float4(vv[0], vv[1], vv[2], vv[3]);
}

// ---------------------------------------------------------------------------------------------
// GENERIC variant (any alignment / any K, N): per-element guarded loads.  Only odd shapes take it
// (e.g. the 1433-wide cora weight); everything on the benchmark path uses gemm_kernel_fast below.
// C[m,n] = epi( sum_k A'[m,k] * B[k,n] ),  B given as W[n,k] (B_KN = false) or W[k,n] (B_KN = true)
// ---------------------------------------------------------------------------------------------
template <int BN, bool B_KN>
__global__ __launch_bounds__(256) void gemm_kernel_generic(const GemmArgs g) {
  constexpr int NT = BN / 64;            // MFMA tiles per wave along N
  constexpr int LDS_N = BN + 4;          // row stride of the [k][n] B tile
  constexpr int A_TILE = BM * LDS_K;
  constexpr int B_TILE = B_KN ? BK * LDS_N : BN * LDS_K;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                      // [2][A_TILE]
  float* Bs = smem + 2 * A_TILE;         // [2][B_TILE]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, kk = lane >> 5;

  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const DropCfg dc = {g.drop_thr, g.drop_seed, g.drop_scale};

  // ---- global->register staging assignment ----
  // A tile: 128 rows x 8 float4; thread f = tid + 256*q -> row f/8, c4 = f%8
  const float* a_ptr[4];
  bool a_ok[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int f = tid + 256 * q;
    const int row = f >> 3;
    const int64_t mrow = m0 + row;
    a_ok[q] = mrow < g.m;
    const int64_t src = a_ok[q] ? (g.a_rows ? g.a_rows[mrow] : mrow) : 0;
    a_ptr[q] = g.a + src * g.lda + (f & 7) * 4;
  }
  constexpr int BQ = B_KN ? (BK * BN / 4) / 256 : (BN * 8) / 256;   // float4 per thread for B

  float4 a_reg[4], b_reg[BQ];

  auto load_tiles = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int kc = k0 + ((tid + 256 * q) & 7) * 4;
      float4 v = a_ok[q] ? load4_guard(a_ptr[q] + k0, g.k - kc, g.a_vec) : zero4();
      if (g.a_scale) v = a_ok[q] ? xform4(v, g.a_scale, g.a_shift, kc, g.k, m0 + ((tid + 256 * q) >> 3), dc) : zero4();
      a_reg[q] = v;
    }
#pragma unroll
    for (int q = 0; q < BQ; ++q) {
      const int f = tid + 256 * q;
      if (B_KN) {
        const int krow = f / (BN / 4), c4 = (f % (BN / 4)) * 4;
        const int kg = k0 + krow, ng = n0 + c4;
        b_reg[q] = (kg < g.k) ? load4_guard(g.b + (int64_t)kg * g.ldb + ng, g.n - ng, g.b_vec) : zero4();
      } else {
        const int nrow = f >> 3, kc = k0 + (f & 7) * 4;
        const int ng = n0 + nrow;
        b_reg[q] = (ng < g.n) ? load4_guard(g.b + (int64_t)ng * g.ldb + kc, g.k - kc, g.b_vec) : zero4();
      }
    }
  };
  auto store_tiles = [&](int buf) {
    float* as = As + buf * A_TILE;
    float* bs = Bs + buf * B_TILE;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int f = tid + 256 * q;
      *reinterpret_cast<float4*>(as + (f >> 3) * LDS_K + (f & 7) * 4) = a_reg[q];
    }
#pragma unroll
    for (int q = 0; q < BQ; ++q) {
      const int f = tid + 256 * q;
      if (B_KN) {
        *reinterpret_cast<float4*>(bs + (f / (BN / 4)) * LDS_N + (f % (BN / 4)) * 4) = b_reg[q];
      } else {
        *reinterpret_cast<float4*>(bs + (f >> 3) * LDS_K + (f & 7) * 4) = b_reg[q];
      }
    }
  };

  f32x16 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (g.k + BK - 1) / BK;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tiles(kt + 1);
    const float* as = As + cur * A_TILE + (wm * 64 + li) * LDS_K + kk * 4;
    const float* bs = B_KN ? Bs + cur * B_TILE + (kk * 4) * LDS_N + wn * (BN / 2) + li
                           : Bs + cur * B_TILE + (wn * (BN / 2) + li) * LDS_K + kk * 4;
#pragma unroll
    for (int kg = 0; kg < BK / 8; ++kg) {
      float af[2][4], bf[NT][4];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(as + i * 32 * LDS_K + kg * 8);
        af[i][0] = v.x; af[i][1] = v.y; af[i][2] = v.z; af[i][3] = v.w;
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (B_KN) {
#pragma unroll
          for (int t = 0; t < 4; ++t) bf[j][t] = bs[(kg * 8 + t) * LDS_N + j * 32];
        } else {
          const float4 v = *reinterpret_cast<const float4*>(bs + j * 32 * LDS_K + kg * 8);
          bf[j][0] = v.x; bf[j][1] = v.y; bf[j][2] = v.z; bf[j][3] = v.w;
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][t], bf[j][t], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tiles(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = n0 + wn * (BN / 2) + j * 32 + li;
    const bool col_ok = col < g.n;
    const float es = (col_ok && g.ep_scale) ? g.ep_scale[col] : 1.f;
    const float eh = (col_ok && g.ep_shift) ? g.ep_shift[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        if (col_ok && row < g.m) {
          float v = acc[i][j][r];
          if (g.row_scale) v *= g.row_scale[row];
          v = fmaf(v, es, eh);
          if (g.relu) v = fmaxf(v, 0.f);
          g.c[row * g.ldc + col] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// FAST variant: every row start 16-byte aligned and every row readable up to round4(extent) (true whenever
// the leading dimension is a multiple of 4).  All global loads are UNCONDITIONAL float4 loads from clamped
// (always valid) addresses, issued back to back before the MFMA block; tails are handled arithmetically:
//   * rows / columns past M / N: the clamped row is re-read (or padding columns are read); their products
//     land in output rows/cols that the epilogue never stores (an output element depends only on its own
//     A row and B row);
//   * k past K: replaced by zeros, per element, on the way to LDS.
// XF: 0 = plain operand, 1 = max(a*scale+shift, 0), 2 = the same + counter-based dropout.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 ld4g(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 mask4(float4 v, int left) {   // keep elements t < left
  v.x = left > 0 ? v.x : 0.f;
  v.y = left > 1 ? v.y : 0.f;
  v.z = left > 2 ? v.z : 0.f;
  v.w = left > 3 ? v.w : 0.f;
  return v;
}

// ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
template <int BMT, int BN, int MI, int NT>
__device__ __forceinline__ void store_tile(const GemmArgs& g, const f32x16 (&acc)[MI][NT], int64_t m0, int n0, int wm, int wn, int li,
                                           int kk) {
  const bool full_tile = (m0 + BMT <= g.m) && (n0 + BN <= g.n);     // workgroup-uniform
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = n0 + wn * (BN / 2) + j * 32 + li;
    const bool col_ok = col < g.n;
    const float es = (col_ok && g.ep_scale) ? g.ep_scale[col] : 1.f;
    const float eh = (col_ok && g.ep_shift) ? g.ep_shift[col] : 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int64_t rbase = m0 + wm * (BMT / 2) + i * 32 + 4 * kk;
      float* cp = g.c + rbase * g.ldc + col;
      float rs[16];
      if (g.row_scale) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int64_t rr = rbase + (r & 3) + 8 * (r >> 2);
          if (rr > g.m - 1) rr = g.m - 1;
          rs[r] = g.row_scale[rr];
        }
      }
      if (full_tile) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[i][j][r];
          if (g.row_scale) v *= rs[r];
          v = fmaf(v, es, eh);
          if (g.relu) v = fmaxf(v, 0.f);
          cp[(int64_t)((r & 3) + 8 * (r >> 2)) * g.ldc] = v;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ro = (r & 3) + 8 * (r >> 2);
          float v = acc[i][j][r];
          if (g.row_scale) v *= rs[r];
          v = fmaf(v, es, eh);
          if (g.relu) v = fmaxf(v, 0.f);
          if (col_ok && rbase + ro < g.m) cp[(int64_t)ro * g.ldc] = v;
        }
      }
    }
  }
}

// BMT x BN block tile, 4 waves as 2 x 2: a wave owns (BMT/2) x (BN/2) = MI x NT MFMA blocks.  BMT = 64 (with BN = 64: one block per
// wave) is the latency form for outputs of a few dozen tiles -- a B=512 student GEMM is 8 tiles of 128x128, i.e. 32 busy SIMDs
// out of 1024 and a serial chain of 64 MFMAs per k-tile and wave; 64x64 tiles quarter that chain and quadruple the workgroups.
template <int BMT, int BN, bool B_KN, int XF, int BKF>
__global__ __launch_bounds__(256) void gemm_kernel_fast(const GemmArgs g) {
  constexpr int MI = BMT / 64;             // 32-row MFMA blocks per wave
  constexpr int LDS_KF = BKF + 4;          // padded [row][k] stride; 9*i / 5*i mod 16 are bijections on the b128 lane groups
  constexpr int KV = BKF / 4;              // float4 per tile row
  constexpr int RSTEP = 256 / KV;          // rows covered by one pass of the 256 threads
  constexpr int AQ = BMT / RSTEP;
  constexpr int NT = BN / 64;
  constexpr int LDS_N = BN + 4;
  constexpr int A_TILE = BMT * LDS_KF;
  constexpr int B_TILE = B_KN ? BKF * LDS_N : BN * LDS_KF;
  constexpr int BQ = B_KN ? (BKF * BN / 4) / 256 : BN / RSTEP;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * A_TILE;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, kk = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * BMT;
  const int n0 = blockIdx.y * BN;
  const int c4 = (tid % KV) * 4;         // k offset of this thread's float4 inside a [row][k] tile
  const int r0 = tid / KV;               // its first row; rows r0 + RSTEP q

  const float* a_base[AQ];
#pragma unroll
  for (int q = 0; q < AQ; ++q) {
    int64_t mrow = m0 + r0 + RSTEP * q;
    if (mrow > g.m - 1) mrow = g.m - 1;
    const int64_t src = g.a_rows ? g.a_rows[mrow] : mrow;
    a_base[q] = g.a + src * g.lda;
  }
  const float* b_base[BQ];
  if (!B_KN) {
#pragma unroll
    for (int q = 0; q < BQ; ++q) {
      int ng = n0 + r0 + RSTEP * q;
      if (ng > g.n - 1) ng = g.n - 1;
      b_base[q] = g.b + (int64_t)ng * g.ldb;
    }
  } else {
    int ng = n0 + (tid % (BN / 4)) * 4;
    const int npad = (g.n + 3) & ~3;
    if (ng > npad - 4) ng = npad - 4;
#pragma unroll
    for (int q = 0; q < BQ; ++q) b_base[q] = g.b + ng;
  }

  const int kpad = (g.k + 3) & ~3;

  // Global -> register -> LDS staging in P = AQ + BQ independent pieces (one float4 per thread each), so that the main
  // loop can place one piece between MFMAs instead of a monolithic load / store phase.
  constexpr int P = AQ + BQ;
  struct Stage {
    float4 a[AQ], b[BQ], sc, sh;
    int kc, k0;       // k of this thread's float4 / of the tile: the store side masks and seeds with them
  };
  Stage st;      // ONE register set: a second set (loads two tiles ahead) made hipcc copy half-tuples at the back-edge
  auto load_prep = [&](Stage& s, int kt) {
    s.k0 = kt * BKF;
    s.kc = s.k0 + c4;
    if (XF) {
      const int kcc = s.kc > kpad - 4 ? kpad - 4 : s.kc;
      s.sc = ld4g(g.a_scale + kcc);
      s.sh = ld4g(g.a_shift + kcc);
    }
  };
  auto load_piece = [&](Stage& s, int p) {
    const int kcc = s.kc > kpad - 4 ? kpad - 4 : s.kc;   // clamped: tiles past K re-read valid addresses, masked at the store
    if (p < AQ) {
      s.a[p] = ld4g(a_base[p] + kcc);
    } else {
      const int q = p - AQ;
      if (B_KN) {
        int kg = s.k0 + (tid + 256 * q) / (BN / 4);
        if (kg > g.k - 1) kg = g.k - 1;
        s.b[q] = ld4g(b_base[q] + (int64_t)kg * g.ldb);
      } else {
        s.b[q] = ld4g(b_base[q] + kcc);
      }
    }
  };
  auto store_piece = [&](const Stage& s, int buf, int p) {
    const int kleft = g.k - s.kc;            // elements t < kleft of this float4 are inside K
    if (p < AQ) {
      float4 v = s.a[p];
      if (XF) {
        v.x = fmaxf(fmaf(v.x, s.sc.x, s.sh.x), 0.f);
        v.y = fmaxf(fmaf(v.y, s.sc.y, s.sh.y), 0.f);
        v.z = fmaxf(fmaf(v.z, s.sc.z, s.sh.z), 0.f);
        v.w = fmaxf(fmaf(v.w, s.sc.w, s.sh.w), 0.f);
        if (XF == 2) {
          const uint32_t row = (uint32_t)(m0 + r0 + RSTEP * p);
          v.x = glnn::drop_keep(g.drop_seed, g.drop_thr, row, (uint32_t)s.kc + 0) ? v.x * g.drop_scale : 0.f;
          v.y = glnn::drop_keep(g.drop_seed, g.drop_thr, row, (uint32_t)s.kc + 1) ? v.y * g.drop_scale : 0.f;
          v.z = glnn::drop_keep(g.drop_seed, g.drop_thr, row, (uint32_t)s.kc + 2) ? v.z * g.drop_scale : 0.f;
          v.w = glnn::drop_keep(g.drop_seed, g.drop_thr, row, (uint32_t)s.kc + 3) ? v.w * g.drop_scale : 0.f;
        }
      }
      v = mask4(v, kleft);
      *reinterpret_cast<float4*>(As + buf * A_TILE + (r0 + RSTEP * p) * LDS_KF + c4) = v;
    } else {
      const int q = p - AQ;
      float4 v = s.b[q];
      float* bs = Bs + buf * B_TILE;
      if (B_KN) {
        const int f = tid + 256 * q;
        const int krow = f / (BN / 4);
        if (s.k0 + krow >= g.k) v = zero4();
        *reinterpret_cast<float4*>(bs + krow * LDS_N + (f % (BN / 4)) * 4) = v;
      } else {
        v = mask4(v, kleft);
        *reinterpret_cast<float4*>(bs + (r0 + RSTEP * q) * LDS_KF + c4) = v;
      }
    }
  };
  auto load_tiles = [&](Stage& s, int kt) {
    load_prep(s, kt);
#pragma unroll
    for (int p = 0; p < P; ++p) load_piece(s, p);
  };

  f32x16 acc[MI][NT];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk_all = (g.k + BKF - 1) / BKF;
  const int kt_beg = blockIdx.z * g.ktiles_per_split;
  int kt_end = kt_beg + g.ktiles_per_split;
  if (kt_end > nk_all) kt_end = nk_all;
  // Main loop, software-pipelined inside every wave (two workgroups per CU run in lockstep, so a wave cannot count on
  // its SIMD neighbour to cover a separate load / ds_write / barrier phase -- measured: 277 us for the bare ds_read+MFMA
  // loop, 296 with a trailing ds_write+barrier phase, 327 with the global loads; hipBLASLt 282 on the same box):
  //   k-groups 0..G-2 : MFMAs of group g while the fragments of group g+1 are read from LDS
  //   then            : ds_write of tile t+1 (global loads issued one whole tile earlier) into the other buffer,
  //                     global loads of tile t+2 issued
  //   last group      : first half of its MFMAs | barrier | ds_read of tile t+1's first fragments | second half
  constexpr int G = BKF / 8;
  float fa[2][MI][4], fb[2][NT][4];
  auto read_frags = [&](int buf, int kg, float (&af)[MI][4], float (&bf)[NT][4]) {
    const float* as = As + buf * A_TILE + (wm * (BMT / 2) + li) * LDS_KF + kk * 4;
    const float* bs = B_KN ? Bs + buf * B_TILE + (kk * 4) * LDS_N + wn * (BN / 2) + li
                           : Bs + buf * B_TILE + (wn * (BN / 2) + li) * LDS_KF + kk * 4;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const float4 v = *reinterpret_cast<const float4*>(as + i * 32 * LDS_KF + kg * 8);
      af[i][0] = v.x; af[i][1] = v.y; af[i][2] = v.z; af[i][3] = v.w;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (B_KN) {
#pragma unroll
        for (int t = 0; t < 4; ++t) bf[j][t] = bs[(kg * 8 + t) * LDS_N + j * 32];
      } else {
        const float4 v = *reinterpret_cast<const float4*>(bs + j * 32 * LDS_KF + kg * 8);
        bf[j][0] = v.x; bf[j][1] = v.y; bf[j][2] = v.z; bf[j][3] = v.w;
      }
    }
  };
  // MFMAs [m_beg, m_end) of one k-group's 4*MI*NT, in the order t (k-pair) -> i (row block) -> j (col block)
  constexpr int MG = 4 * MI * NT;
  auto mfma_range = [&](const float (&af)[MI][4], const float (&bf)[NT][4], int m_beg, int m_end) {
#pragma unroll
    for (int mm = m_beg; mm < m_end; ++mm) {
      const int t = mm / (MI * NT), i = (mm / NT) % MI, j = mm % NT;
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][t], bf[j][t], acc[i][j], 0, 0, 0);
    }
  };

  load_tiles(st, kt_beg);
#pragma unroll
  for (int p = 0; p < P; ++p) store_piece(st, 0, p);
  __syncthreads();
  read_frags(0, 0, fa[0], fb[0]);
  load_tiles(st, kt_beg + 1);

  for (int kt = kt_beg; kt < kt_end; ++kt) {
    const int cur = (kt - kt_beg) & 1;
    // groups 0 .. G-3: MFMAs while the next group's fragments are read
#pragma unroll
    for (int kg = 0; kg + 2 < G; ++kg) {
      read_frags(cur, kg + 1, fa[(kg + 1) & 1], fb[(kg + 1) & 1]);
      mfma_range(fa[kg & 1], fb[kg & 1], 0, MG);
    }
    // group G-2: + tile kt+1 (loads issued most of a tile ago) goes registers -> the other LDS buffer, one piece per
    // MG/P MFMAs.  Past the last tile this writes a buffer nobody reads and the loads below re-read clamped, in-bounds
    // addresses: unconditional, so that the loop body stays one basic block.
    read_frags(cur, G - 1, fa[(G - 1) & 1], fb[(G - 1) & 1]);
#pragma unroll
    for (int p = 0; p < P; ++p) {
      __builtin_amdgcn_sched_barrier(0);
      store_piece(st, cur ^ 1, p);
      __builtin_amdgcn_sched_barrier(0);
      mfma_range(fa[(G - 2) & 1], fb[(G - 2) & 1], p * MG / P, (p + 1) * MG / P);
    }
    // group G-1, first half: + the freed registers are re-loaded with tile kt+2, one piece per MFMA slot
    load_prep(st, kt + 2);
#pragma unroll
    for (int p = 0; p < P; ++p) {
      __builtin_amdgcn_sched_barrier(0);
      load_piece(st, p);
      __builtin_amdgcn_sched_barrier(0);
      mfma_range(fa[(G - 1) & 1], fb[(G - 1) & 1], p * (MG / 2) / P, (p + 1) * (MG / 2) / P);
    }
    // barrier | the next tile's first fragment reads | second half of group G-1
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    read_frags(cur ^ 1, 0, fa[0], fb[0]);
    __builtin_amdgcn_sched_barrier(0);
    mfma_range(fa[(G - 1) & 1], fb[(G - 1) & 1], MG / 2, MG);
  }

  if (g.ksplits > 1) {
    // split-K: raw partial sums to the workspace slab of this k-range; the reduce kernel applies the epilogue
    float* wsz = g.ws + (int64_t)blockIdx.z * g.m * g.n;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = n0 + wn * (BN / 2) + j * 32 + li;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int64_t rbase = m0 + wm * (BMT / 2) + i * 32 + 4 * kk;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = rbase + (r & 3) + 8 * (r >> 2);
          if (col < g.n && row < g.m) wsz[row * g.n + col] = acc[i][j][r];
        }
      }
    }
    return;
  }

  store_tile<BMT, BN, MI, NT>(g, acc, m0, n0, wm, wn, li, kk);
}

// ---------------------------------------------------------------------------------------------
// PIPE variant of the 128 x 128 x 32 kernels (plain operands, K a multiple of 32): the same tiles, k pairing and MFMA
// order as gemm_kernel_fast / gemm_tn_kernel_t -- results are bit-identical -- but the steady-state loop is stated
// instruction by instruction (asm volatile, hand-counted s_waitcnt) and contains NO vector-ALU work at all:
//   * placement: hipcc treats the MFMA builtins as pure values and lets them float past sched_barrier, which left the 8
//     ds_write_b128 of a k-tile back to back and its 8 global loads in one burst; here every 8th MFMA slot carries ONE
//     {s_waitcnt vmcnt(7) -> ds_write_b128 -> buffer_load_dwordx4 into the same registers} piece, so a load has a whole
//     k-tile (64 MFMAs of this wave) to arrive;
//   * no VALU: a register-only MFMA loop on this part runs at 155 TF; with the GEMM's LDS/global instruction mix between the
//     MFMAs 142-147 TF (experiments/mfma_mix.hip); every additional VALU instruction (64-bit address adds, k-tail selects)
//     competes with the MFMAs for the SIMD's issue port.  Operands are therefore addressed through buffer descriptors
//     (per-thread byte offsets fixed before the loop, the k-tile step in one SGPR), both LDS buffers through immediate
//     offsets (loop unrolled by two k-tiles), and the k tail is not handled here at all (K % 32 != 0 takes gemm_kernel_fast).
//   per k-group kg (16 MFMAs): m0 m1 m2 [piece 2kg] m3 m4 r m5 r m6 r m7 r m8 m9 [piece 2kg+1] m10 .. m15     (r = ds_read_b128 of
//   group kg+1); in the last group the barrier follows m11, then the 4 reads of the next tile's first group, then m12..m15.
// The compiler still allocates the registers.
// ---------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// (accumulators in arch VGPRs: the blocked accumulation below adds them up with VALU instructions; AGPR accumulators + accvgpr
//  reads / writes measured 1.2 % slower on the 4096 x 2048 x 2048 products)
#define GLNN_MFMA(ACC_, A_, B_) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(ACC_) : "v"(A_), "v"(B_))
#define GLNN_DS_READ(DST_, ADDR_, OFF_) \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST_) : "v"(ADDR_), "n"(OFF_) : "memory")
#define GLNN_DS_READ2ST64(DST_, ADDR_, OFF0_, OFF1_) \
  asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(DST_) : "v"(ADDR_), "n"(OFF0_), "n"(OFF1_) : "memory")
#define GLNN_DS_WRITE(ADDR_, SRC_, OFF_) \
  asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(ADDR_), "v"(SRC_), "n"(OFF_) : "memory")
#define GLNN_BLOAD(DST_, VOFF_, RSRC_, SOFF_) \
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(DST_) : "v"(VOFF_), "s"(RSRC_), "s"(SOFF_) : "memory")

// raw buffer descriptor over [base, base + bytes): stride 0, 32-bit data format.  A load whose offset (register + SGPR
// step) reaches num_records returns 0 instead of touching memory: k rows past the end of a KROW operand (the last, partial
// k-tile of a weight-gradient split; the prefetches past the last tile) need neither a clamp nor a mask.
__device__ __forceinline__ i32x4 make_rsrc(const float* base, int64_t bytes) {
  const uint64_t b = reinterpret_cast<uint64_t>(base);
  i32x4 r;
  r.x = (int)(uint32_t)b;
  r.y = (int)(uint32_t)((b >> 32) & 0xFFFFu);
  r.z = (int)(bytes < 0 ? 0 : (bytes > 0x7FFFFFFF ? 0x7FFFFFFF : bytes));
  r.w = 0x00020000;
  return r;
}

// How a 128 x 32 operand tile lives in memory and in LDS:
//   ROWK  rows = the output index, k contiguous (A of every forward GEMM, W[n,k]): LDS [row][k] padded to 36 floats, one
//         ds_read_b128 per 32-row block and k-group (lane half kk takes k = 8 kg + 4 kk .. +3);
//   KROW  rows = k, the output index contiguous (W[k,n] of the input-gradient GEMM, both operands of the weight-gradient
//         GEMM): LDS [k][128] unpadded (a fragment read touches 32 consecutive floats of one k row), two
//         ds_read2st64_b32 per block and k-group -- the same k per lane and MFMA as ROWK, without any transpose.
enum { ROWK = 0, KROW = 1 };
template <int STYLE> struct PipeOp;
template <> struct PipeOp<ROWK> {
  static constexpr int TILE = 128 * 36 * 4;      // bytes per LDS buffer
  static constexpr int PIECE = 32 * 36 * 4;      // byte step between a thread's four staging pieces
  static constexpr int READS = 2;                // DS reads per k-group (both 32-row blocks)
  f32x4 q[2];
};
template <> struct PipeOp<KROW> {
  static constexpr int TILE = 32 * 128 * 4;
  static constexpr int PIECE = 8 * 128 * 4;
  static constexpr int READS = 4;
  f32x2 h[2][2];
};

// acc += A_tile . B_tile^T over nk k-tiles of 32 for ONE 128 x 128 output tile (4 waves as 2 x 2, 2 x 2 MFMA blocks each).
//   a0 / b0   the tile's origin: ROWK -> element (first row, first k), KROW -> element (first k, first column)
//   ext_a/b   valid rows (ROWK) / valid columns rounded up to 4 (KROW) from the origin: further ones are clamped (they only
//             feed outputs that are never stored)
//   kext      k extent from the origin.  ROWK operands need kext == 32 nk (a k tail inside a row is not out of range);
//             KROW operands may end inside the last k-tile: those rows lie behind the descriptor's end and read as 0
#ifndef GLNN_PIPE_BLOCK_TILES
#define GLNN_PIPE_BLOCK_TILES 8
#endif
constexpr int PIPE_BLOCK_TILES = GLNN_PIPE_BLOCK_TILES;      // k-tiles (of 32) per accumulation block: a power of two >= 2
// AX (KROW A only, round 5): the A operand is alpha[col] * a + beta[col] * z + gamma[col] with a second matrix z of A's shape and per-column
// constants -- the BatchNorm backward's dz written as an affine map of (dy, z), glnn::BnApplyA -- evaluated on the staged piece between its
// s_waitcnt and its ds_write_b128 (two v_pk_fma_f32 pairs per piece: the only VALU work of the loop); z pieces ride behind the A pieces
// in the load order (12 loads per k-tile in flight instead of 8).  Rows behind the descriptors' end are gamma, against B rows that are 0.
struct PipeAx { const float* z0; int64_t ldz; const float* alpha; const float* beta; const float* gamma;
                // (round 6) alpha == NULL: the constants are made HERE from the still unfolded tile partials S1 / S2 (p1 / p2 [nparts][h], already
                // offset to the tile's first column) -- bn_bwd_parts_consts_kernel's arithmetic and order, one launch less in front of the product
                const float* p1; const float* p2; int nparts; int64_t h; float inv_rows; const float* bn_gamma; const float* bn_mean; const float* bn_rstd;
                float* dgamma; float* dbeta; float* colsum; };
template <int SA, int SB, bool AX = false>
__device__ __forceinline__ void pipe_mainloop(const float* a0, int64_t lda, int64_t ext_a, const float* b0, int64_t ldb, int64_t ext_b,
                                              int64_t kext, int nk, f32x16 (&acc)[2][2], const PipeAx* ax = nullptr) {
  static_assert(!AX || SA == KROW, "the operand transform is stated for the k-major A of the weight-gradient product");
  using OA = PipeOp<SA>;
  using OB = PipeOp<SB>;
  constexpr int A_OFF = 0, B_OFF = 2 * OA::TILE;
  constexpr int NRA = OA::READS, NR = OA::READS + OB::READS;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, kk = lane >> 5;

  const i32x4 rsrc_a = make_rsrc(a0, SA == ROWK ? ((ext_a - 1) * lda + kext) * 4 : ((kext - 1) * lda + ext_a) * 4);
  const i32x4 rsrc_b = make_rsrc(b0, SB == ROWK ? ((ext_b - 1) * ldb + kext) * 4 : ((kext - 1) * ldb + ext_b) * 4);
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;       // the dynamic array is the kernel's only LDS object
  uint32_t voff[8];                      // per-piece global byte offsets: 0..3 = A, 4..7 = B
  uint32_t wr_a, wr_b;                   // LDS write bases (buffer 0)
  uint32_t rd_a[2], rd_b[2];             // LDS fragment read bases (ROWK uses [0] only)
  uint32_t step_a, step_b;               // global byte step per k-tile
  auto setup = [&](auto style_, const int64_t ld, int64_t ext, int w, uint32_t lds_base, uint32_t* vo, uint32_t& wr, uint32_t (&rd)[2],
                   uint32_t& step) {
    constexpr int ST = decltype(style_)::value;
    if constexpr (ST == ROWK) {
      const int c4 = (tid & 7) * 4, r0 = tid >> 3;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int64_t r = r0 + 32 * q;
        if (r > ext - 1) r = ext - 1;
        vo[q] = (uint32_t)((r * ld + c4) * 4);
      }
      wr = lds_base + (uint32_t)((r0 * 36 + c4) * 4);
      rd[0] = rd[1] = lds_base + (uint32_t)(((w * 64 + li) * 36 + kk * 4) * 4);
      step = 32 * 4;
    } else {
      const int kr = tid >> 5;
      int64_t n4 = (tid & 31) * 4;
      if (n4 > ext - 4) n4 = ext - 4;
#pragma unroll
      for (int q = 0; q < 4; ++q) vo[q] = (uint32_t)(((kr + 8 * q) * ld + n4) * 4);
      wr = lds_base + (uint32_t)((kr * 128 + (tid & 31) * 4) * 4);
      rd[0] = lds_base + (uint32_t)(((kk * 4) * 128 + w * 64 + li) * 4);
      rd[1] = rd[0] + 32 * 4;
      step = (uint32_t)(32 * ld * 4);
    }
  };
  setup(std::integral_constant<int, SA>{}, lda, ext_a, wm, lds0 + A_OFF, voff, wr_a, rd_a, step_a);
  setup(std::integral_constant<int, SB>{}, ldb, ext_b, wn, lds0 + B_OFF, voff + 4, wr_b, rd_b, step_b);

  f32x4 st[8];                           // the staged tile: one float4 per piece
  // AX: the z pieces next to the A pieces, their descriptor / offsets / step, and the lane's per-column constants
  f32x4 zt[4];
  i32x4 rsrc_z = rsrc_a;
  uint32_t voff_z[4] = {0, 0, 0, 0}, step_z = 0;
  f32x4 ax_al = {0.f, 0.f, 0.f, 0.f}, ax_be = ax_al, ax_ga = ax_al;
  if constexpr (AX) {
    const int kr = tid >> 5;
    int64_t n4 = (tid & 31) * 4;
    if (n4 > ext_a - 4) n4 = ext_a - 4;
    rsrc_z = make_rsrc(ax->z0, ((kext - 1) * ax->ldz + ext_a) * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) voff_z[q] = (uint32_t)(((kr + 8 * q) * ax->ldz + n4) * 4);
    step_z = (uint32_t)(32 * ax->ldz * 4);
    if (ax->alpha) {
      ax_al = *reinterpret_cast<const f32x4*>(ax->alpha + n4);
      ax_be = *reinterpret_cast<const f32x4*>(ax->beta + n4);
      ax_ga = *reinterpret_cast<const f32x4*>(ax->gamma + n4);
    } else {
      f32x4 S1 = {0.f, 0.f, 0.f, 0.f}, S2 = S1;
      for (int k0 = 0; k0 < ax->nparts; k0 += 16) {
        f32x4 t1[16], t2[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int k = k0 + u < ax->nparts ? k0 + u : ax->nparts - 1;
          t1[u] = *reinterpret_cast<const f32x4*>(ax->p1 + (int64_t)k * ax->h + n4);
          t2[u] = *reinterpret_cast<const f32x4*>(ax->p2 + (int64_t)k * ax->h + n4);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
          if (k0 + u < ax->nparts) { S1 += t1[u]; S2 += t2[u]; }
      }
      const f32x4 rs = *reinterpret_cast<const f32x4*>(ax->bn_rstd + n4), gm = *reinterpret_cast<const f32x4*>(ax->bn_gamma + n4);
      const f32x4 mu = *reinterpret_cast<const f32x4*>(ax->bn_mean + n4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float c1 = S1[i] * ax->inv_rows, c2 = S2[i] * ax->inv_rows, grs = gm[i] * rs[i];
        const float kq = grs * rs[i] * c2;
        ax_al[i] = grs;
        ax_be[i] = -kq;
        ax_ga[i] = fmaf(kq, mu[i], -grs * c1);
      }
      if (ax->dgamma && tid < 32) {          // one workgroup per column block writes the parameter gradients (the caller passes dgamma to those only)
        *reinterpret_cast<f32x4*>(ax->dbeta + n4) = S1;
        *reinterpret_cast<f32x4*>(ax->dgamma + n4) = S2;
        if (ax->colsum) *reinterpret_cast<f32x4*>(ax->colsum + n4) = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
  auto soff_z = [&](int kt) { return kt < nk ? (uint32_t)kt * step_z : (uint32_t)rsrc_z.z; };
  // SGPR offset of k-tile kt (uniform: SALU).  A tile past the end gets the descriptor's own size as offset: every lane is out of
  // range and reads 0 without touching memory -- the prefetches behind the last tile, and the zero tile that pads an odd tile count
  auto soff_a = [&](int kt) { return kt < nk ? (uint32_t)kt * step_a : (uint32_t)rsrc_a.z; };
  auto soff_b = [&](int kt) { return kt < nk ? (uint32_t)kt * step_b : (uint32_t)rsrc_b.z; };

  // ---- prologue: tile 0 -> LDS buffer 0 ----
  {
    const uint32_t sa = 0, sb = 0;
    static_for<8>([&](auto p_) {
      constexpr int p = decltype(p_)::value;
      (void)st; (void)voff; (void)rsrc_a; (void)rsrc_b; (void)sa; (void)sb;   // asm-only operands are not captured implicitly (clang)
      (void)zt; (void)voff_z; (void)rsrc_z;
      if constexpr (p < 4) {
        GLNN_BLOAD(st[p], voff[p], rsrc_a, sa);
        if constexpr (AX) GLNN_BLOAD(zt[p], voff_z[p], rsrc_z, sa);
      } else GLNN_BLOAD(st[p], voff[p], rsrc_b, sb);
    });
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(st[0]), "+v"(st[1]), "+v"(st[2]), "+v"(st[3]), "+v"(st[4]), "+v"(st[5]), "+v"(st[6]), "+v"(st[7]) : : "memory");
    if constexpr (AX) {
      asm volatile("" : "+v"(zt[0]), "+v"(zt[1]), "+v"(zt[2]), "+v"(zt[3]) : : "memory");      // (landed with the wait above)
#pragma unroll
      for (int p = 0; p < 4; ++p) st[p] = st[p] * ax_al + (zt[p] * ax_be + ax_ga);
    }
    static_for<8>([&](auto p_) {
      constexpr int p = decltype(p_)::value;
      (void)wr_a; (void)wr_b; (void)st;
      if constexpr (p < 4) GLNN_DS_WRITE(wr_a, st[p], p * OA::PIECE);
      else GLNN_DS_WRITE(wr_b, st[p], (p - 4) * OB::PIECE);
    });
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  PipeOp<SA> fa[2];                      // fragments, double-buffered by k-group parity
  PipeOp<SB> fb[2];
  f32x16 tot[2][2];                      // sum of the finished blocks (see kloop)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.f;

  // the DS reads of k-group KG out of LDS buffer BUF into fragment set PAR; R = 0 .. NR-1 selects one instruction
  auto frag_read = [&](auto buf_, auto kg_, auto par_, auto r_) {
    constexpr int BUF = decltype(buf_)::value, KG = decltype(kg_)::value, PAR = decltype(par_)::value, R = decltype(r_)::value;
    (void)fa; (void)fb; (void)rd_a; (void)rd_b;
    if constexpr (R < NRA) {
      if constexpr (SA == ROWK) GLNN_DS_READ(fa[PAR].q[R], rd_a[0], BUF * OA::TILE + R * OA::PIECE + KG * 32);
      else GLNN_DS_READ2ST64(fa[PAR].h[R / 2][R % 2], rd_a[R / 2], BUF * 64 + 2 * (KG * 8 + 2 * (R % 2)), BUF * 64 + 2 * (KG * 8 + 2 * (R % 2) + 1));
    } else {
      constexpr int Q = R - NRA;
      if constexpr (SB == ROWK) GLNN_DS_READ(fb[PAR].q[Q], rd_b[0], BUF * OB::TILE + Q * OB::PIECE + KG * 32);
      else GLNN_DS_READ2ST64(fb[PAR].h[Q / 2][Q % 2], rd_b[Q / 2], BUF * 64 + 2 * (KG * 8 + 2 * (Q % 2)), BUF * 64 + 2 * (KG * 8 + 2 * (Q % 2) + 1));
    }
  };

  // one k-tile out of LDS buffer CUR; tile kt+1 goes registers -> buffer CUR^1, tile kt+2 global -> registers.  The two
  // staging pieces of a k-group sit behind MFMA slots PH and PH+7 (giving every wave of the workgroup its own PH -- four copies
  // of the loop -- so that the waves do not hand their ds_write_b128 to the LDS store path in the same slot measured equal).
  auto ktile = [&](auto cur_, auto ph_, uint32_t sa, uint32_t sb, uint32_t sz) {
    constexpr int CUR = decltype(cur_)::value, PH = decltype(ph_)::value;
    static_for<64>([&](auto m_) {
      constexpr int kg = decltype(m_)::value / 16, mm = decltype(m_)::value % 16;
      constexpr int par = kg & 1;
      (void)wr_a; (void)wr_b; (void)sa; (void)sb; (void)st; (void)voff; (void)rsrc_a; (void)rsrc_b; (void)fa; (void)fb; (void)acc;
      (void)sz; (void)zt; (void)voff_z; (void)rsrc_z; (void)ax_al; (void)ax_be; (void)ax_ga;
      // ^ operands that appear only inside asm / discarded branches are not captured implicitly (clang)
      if constexpr (mm == 0) {
        // fragments of this group: issued >= 4 MFMAs ago; the previous group's second ds_write follows them only if its slot
        // (PH + 7) lies behind the last read (slot 3 + NR)
        if constexpr (kg == 0 || PH + 7 <= 3 + NR) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
      }
      constexpr int t = mm / 4, i = (mm / 2) % 2, j = mm % 2;
      if constexpr (SA == ROWK && SB == ROWK) GLNN_MFMA(acc[i][j], fa[par].q[i][t], fb[par].q[j][t]);
      if constexpr (SA == ROWK && SB == KROW) GLNN_MFMA(acc[i][j], fa[par].q[i][t], fb[par].h[j][t / 2][t % 2]);
      if constexpr (SA == KROW && SB == KROW) GLNN_MFMA(acc[i][j], fa[par].h[i][t / 2][t % 2], fb[par].h[j][t / 2][t % 2]);
      if constexpr (SA == KROW && SB == ROWK) GLNN_MFMA(acc[i][j], fa[par].h[i][t / 2][t % 2], fb[par].q[j][t]);
      if constexpr (mm == PH || mm == PH + 7) {
        constexpr int p = 2 * kg + (mm == PH + 7 ? 1 : 0);
        // piece p: the load issued one k-tile ago has 7 younger ones behind it (AX: an A piece and its z piece have 10 behind them, a B piece 11)
        if constexpr (!AX) asm volatile("s_waitcnt vmcnt(7)" : "+v"(st[p]) : : "memory");
        else if constexpr (p < 4) asm volatile("s_waitcnt vmcnt(10)" : "+v"(st[p]), "+v"(zt[p]) : : "memory");
        else asm volatile("s_waitcnt vmcnt(11)" : "+v"(st[p]) : : "memory");
        if constexpr (p < 4) {
          if constexpr (AX) st[p] = st[p] * ax_al + (zt[p] * ax_be + ax_ga);
          GLNN_DS_WRITE(wr_a, st[p], (CUR ^ 1) * OA::TILE + p * OA::PIECE);
          GLNN_BLOAD(st[p], voff[p], rsrc_a, sa);
          if constexpr (AX) GLNN_BLOAD(zt[p], voff_z[p], rsrc_z, sz);
        } else {
          GLNN_DS_WRITE(wr_b, st[p], (CUR ^ 1) * OB::TILE + (p - 4) * OB::PIECE);
          GLNN_BLOAD(st[p], voff[p], rsrc_b, sb);
        }
      }
      if constexpr (kg < 3 && mm >= 4 && mm < 4 + NR)        // fragments of group kg+1 (other parity), one instruction per slot
        frag_read(cur_, std::integral_constant<int, kg + 1>{}, std::integral_constant<int, par ^ 1>{}, std::integral_constant<int, mm - 4>{});
      if constexpr (kg == 3 && mm == 11) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        static_for<NR>([&](auto r_) {
          frag_read(std::integral_constant<int, CUR ^ 1>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, r_);
        });
      }
    });
  };
  // everything that is in flight when the loop starts (tile 1, the first fragments) is issued right in front of it: the compiler
  // cannot see that an asm load has not landed yet, so the loaded values must not cross a branch where it could copy them
  auto kloop = [&](auto ph_) {
    {
      const uint32_t sa = soff_a(1), sb = soff_b(1), sz = soff_z(1);
      static_for<8>([&](auto p_) {
        constexpr int p = decltype(p_)::value;
        (void)st; (void)voff; (void)rsrc_a; (void)rsrc_b; (void)sa; (void)sb; (void)sz; (void)zt; (void)voff_z; (void)rsrc_z;
        if constexpr (p < 4) {
          GLNN_BLOAD(st[p], voff[p], rsrc_a, sa);
          if constexpr (AX) GLNN_BLOAD(zt[p], voff_z[p], rsrc_z, sz);
        } else GLNN_BLOAD(st[p], voff[p], rsrc_b, sb);
      });
    }
    static_for<NR>([&](auto r_) {
      frag_read(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, r_);
    });
    // always an even number of k-tiles (an odd count is padded with one all-zero tile): a separate tail after the loop is a
    // control-flow join, where the compiler may copy the staged registers -- while their asm loads are still in flight
    for (int kt = 0; kt < nk; kt += 2) {
      ktile(std::integral_constant<int, 0>{}, ph_, soff_a(kt + 2), soff_b(kt + 2), soff_z(kt + 2));
      ktile(std::integral_constant<int, 1>{}, ph_, soff_a(kt + 3), soff_b(kt + 3), soff_z(kt + 3));
      // blocked accumulation: the MFMA chain of an output element is cut every PIPE_BLOCK_TILES k-tiles (256 k) and its partial sum
      // moved into `tot` by VALU adds -- 32 v_pk_add + 32 v_mov_b64 per wave and block.  fp32 rounding noise of a 4096 x K x 2048
      // product against fp64 (scripts/gemm_noise.py, rms relative): one chain 5.7e-7 / 8.1e-7 / 1.15e-6 at K = 1024 / 2048 / 4096
      // (= rocBLAS), blocked 2.9e-7 at every K (numpy's blocked sgemm: 3.4 - 3.8e-7), for +0.3 .. 2 % time.  A uniform, rarely
      // taken branch; nothing but acc / tot is touched inside (the staged loads and fragment reads in flight keep their registers)
      if (__builtin_expect(((kt + 2) & (PIPE_BLOCK_TILES - 1)) == 0 && kt + 2 < nk, 0)) {
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));   // XDL write -> VALU read: 18 wait states
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            tot[i][j] += acc[i][j];
            acc[i][j] = 0.f;
          }
        asm volatile("s_nop 4" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));                // VALU write -> XDL SrcC read
      }
    }
    // drain: loads / fragment reads of the tiles past the end are in flight; the accumulators are read by VALU next (XDL
    // write -> VALU read needs 18 wait states the compiler cannot see behind the asm)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = tot[i][j] + acc[i][j];
  };
  kloop(std::integral_constant<int, 2>{});
}

// GLNN_GEMM_PIPE=0 keeps every shape on the compiler-scheduled kernels (A/B runs, tests/test_kernels_gpu.py)
inline bool pipe_enabled() { return glnn::opts().gemm_pipe != 0; }

// GLNN_GEMM_ROWPANEL=0 keeps short reductions on the tiled kernels (A/B runs, bit-identity tests)
inline bool rowpanel_enabled() { return glnn::opts().gemm_rowpanel != 0; }

constexpr size_t pipe_lds_bytes(int sa, int sb) {
  return 2 * (size_t)((sa == ROWK ? PipeOp<ROWK>::TILE : PipeOp<KROW>::TILE) + (sb == ROWK ? PipeOp<ROWK>::TILE : PipeOp<KROW>::TILE));
}

// Column statistics of one stored 128 x 128 tile (g.st_mean != NULL; no row scale, no relu): mean and M2 = sum (v - mean)^2 over the
// tile's valid rows, two passes over the accumulators like bn_stats_stage1's two passes over memory, written for the tile row
// m0 / 128.  A column's 128 values sit in 2 lanes (lane halves) x 2 waves (row halves) x 32 registers: shuffle, then LDS.
__device__ __forceinline__ void pipe_tile_stats(const GemmArgs& g, const f32x16 (&acc)[2][2], int64_t m0, int n0, int wm, int wn, int li, int kk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];      // free: every LDS write of the main loop is behind its last barrier
  float* red = smem;                                                // [2 row halves][128 columns]
  const int rows = (int)(g.m - m0 < 128 ? g.m - m0 : 128);
  float v[2][2][16];
  float s[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn * 64 + j * 32 + li;
    const bool col_ok = col < g.n;
    const float es = (col_ok && g.ep_scale) ? g.ep_scale[col] : 1.f;
    const float eh = (col_ok && g.ep_shift) ? g.ep_shift[col] : 0.f;
    s[j] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + i * 32 + 4 * kk + (r & 3) + 8 * (r >> 2);
        v[j][i][r] = fmaf(acc[i][j][r], es, eh);                    // the stored value
        s[j] += row < rows ? v[j][i][r] : 0.f;
      }
    s[j] += __shfl_xor(s[j], 32);
  }
  __syncthreads();
  if (kk == 0) { red[wm * 128 + wn * 64 + li] = s[0]; red[wm * 128 + wn * 64 + 32 + li] = s[1]; }
  __syncthreads();
  float mean[2], q[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = wn * 64 + j * 32 + li;
    mean[j] = (red[c] + red[128 + c]) / (float)rows;
    q[j] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + i * 32 + 4 * kk + (r & 3) + 8 * (r >> 2);
        const float dv = v[j][i][r] - mean[j];
        q[j] = row < rows ? fmaf(dv, dv, q[j]) : q[j];
      }
    q[j] += __shfl_xor(q[j], 32);
  }
  __syncthreads();
  if (kk == 0) { red[wm * 128 + wn * 64 + li] = q[0]; red[wm * 128 + wn * 64 + 32 + li] = q[1]; }
  __syncthreads();
  if (wm == 0 && kk == 0) {
    const int64_t base = (m0 / 128) * (int64_t)g.n;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = wn * 64 + j * 32 + li, col = n0 + c;
      if (col < g.n) {
        g.st_mean[base + col] = mean[j];
        g.st_m2[base + col] = red[c] + red[128 + c];
      }
    }
  }
}

// g.bd_z != NULL: acc (= da of the tile) -> dy in place, column partial sums of the tile's valid rows to bd_s1 / bd_s2.  The element
// arithmetic is bn_dy's (student.hip): dropout decision by glnn::drop_keep's hash, ReLU gate on fma(z, a_scale, a_shift) > 0.  A column's 128
// values sit in 2 lanes (lane halves) x 2 waves (row halves) x 32 registers: registers ascending, shuffle, then LDS -- a fixed order.
__device__ __forceinline__ void pipe_tile_bn_dy(const GemmArgs& g, f32x16 (&acc)[2][2], int64_t m0, int n0, int wm, int wn, int li, int kk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];      // free: every LDS write of the main loop is behind its last barrier
  float* red1 = smem;                                               // [2 row halves][128 columns]
  float* red2 = smem + 256;
  const int rows = (int)(g.m - m0 < 128 ? g.m - m0 : 128);
  float zz[2][2][16];
  int colc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn * 64 + j * 32 + li;
    colc[j] = col < g.n ? col : g.n - 1;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {                                // all 64 loads of the lane are in flight together
        const int row = wm * 64 + i * 32 + 4 * kk + (r & 3) + 8 * (r >> 2);
        const int64_t gr = m0 + (row < rows ? row : rows - 1);
        zz[j][i][r] = g.bd_z[gr * g.bd_ldz + colc[j]];
      }
  }
  float s1[2], s2[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float mu = g.bd_mean[colc[j]], rs = g.bd_rstd[colc[j]], sc = g.bd_scale[colc[j]], sf = g.bd_shift[colc[j]];
    const uint32_t hcol = g.bd_seed ^ (((uint32_t)colc[j] >> 1) * 0x85EBCA77u + 0x632BE5ABu);
    const bool hi = colc[j] & 1;
    s1[j] = 0.f; s2[j] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + i * 32 + 4 * kk + (r & 3) + 8 * (r >> 2);
        float dav = acc[i][j][r];
        if (g.bd_thr) {
          uint32_t h = hcol ^ ((uint32_t)(m0 + row) * 0x9E3779B1u);
          h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
          dav = (hi ? (h >> 16) : (h & 0xFFFFu)) >= g.bd_thr ? dav * g.bd_dscale : 0.f;
        }
        const bool on = !g.bd_relu || fmaf(zz[j][i][r], sc, sf) > 0.f;
        const float dy = (on && row < rows) ? dav : 0.f;
        acc[i][j][r] = dy;
        s1[j] += dy;
        s2[j] = fmaf(dy, (zz[j][i][r] - mu) * rs, s2[j]);
      }
    s1[j] += __shfl_xor(s1[j], 32);
    s2[j] += __shfl_xor(s2[j], 32);
  }
  __syncthreads();
  if (kk == 0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) { red1[wm * 128 + wn * 64 + j * 32 + li] = s1[j]; red2[wm * 128 + wn * 64 + j * 32 + li] = s2[j]; }
  }
  __syncthreads();
  if (wm == 0 && kk == 0) {
    const int64_t base = (m0 / 128) * (int64_t)g.n;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = wn * 64 + j * 32 + li, col = n0 + c;
      if (col < g.n) {
        g.bd_s1[base + col] = red1[c] + red1[128 + c];
        g.bd_s2[base + col] = red2[c] + red2[128 + c];
      }
    }
  }
}

// C = epi(A . W^T) (B_KN = false, W [n,k]) or epi(A . W) (B_KN = true, W [k,n]): plain A, K % 32 == 0, no split-K
template <bool B_KN>
__global__ __launch_bounds__(256) void gemm_kernel_pipe(const GemmArgs g) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t m0 = (int64_t)blockIdx.x * 128;
  const int n0 = blockIdx.y * 128;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if (B_KN)
    pipe_mainloop<ROWK, KROW>(g.a + m0 * g.lda, g.lda, g.m - m0, g.b + n0, g.ldb, ((g.n + 3) & ~3) - n0, g.k, g.k / BK, acc);
  else
    pipe_mainloop<ROWK, ROWK>(g.a + m0 * g.lda, g.lda, g.m - m0, g.b + (int64_t)n0 * g.ldb, g.ldb, g.n - n0, g.k, g.k / BK, acc);
  if constexpr (B_KN) {
    if (g.bd_z) pipe_tile_bn_dy(g, acc, m0, n0, wave >> 1, wave & 1, lane & 31, lane >> 5);
  }
  store_tile<128, 128, 2, 2>(g, acc, m0, n0, wave >> 1, wave & 1, lane & 31, lane >> 5);
  if (g.st_mean) pipe_tile_stats(g, acc, m0, n0, wave >> 1, wave & 1, lane & 31, lane >> 5);
}

// ---------------------------------------------------------------------------------------------
// TN: C[i,j] = sum_m A'[m,i] * B[m,j]   (i < ka, j < nb), reduction over rows m, optional split over m.
// Both operand tiles are [m][*] row-major in global and stay k(m)-major in LDS ([BK][128+4]); MFMA
// fragments are ds_read_b32 with consecutive lanes on consecutive floats.
// ---------------------------------------------------------------------------------------------
struct GemmTnArgs {
  const float* a; int64_t lda;
  int64_t m; int ka;
  const float* b; int64_t ldb; const int64_t* b_rows; const float* b_scale; const float* b_shift; int nb;
  float* c; int64_t ldc;   // final C, or the split workspace [splits][ka][nb] when splits > 1
  int splits; int64_t rows_per_split;
  int a_vec; int b_vec;
  uint32_t drop_thr; uint32_t drop_seed; float drop_scale;
  // gemm_tn_kernel_pipe_ax only: A = ax_alpha * a + ax_beta * ax_z + ax_gamma per column (glnn::BnApplyA)
  const float* ax_z; int64_t ax_ldz; const float* ax_alpha; const float* ax_beta; const float* ax_gamma;
  // ... or (ax_alpha == NULL) the constants made in the kernel's prologue from the tile partials (PipeAx)
  const float* ax_p1; const float* ax_p2; int ax_nparts; float ax_inv_rows; const float* ax_bn_gamma; const float* ax_bn_mean; const float* ax_bn_rstd;
  float* ax_dgamma; float* ax_dbeta; float* ax_colsum;
};

template <int BNT>
__global__ __launch_bounds__(256) void gemm_tn_kernel_generic(const GemmTnArgs g) {
  // GENERIC variant (any alignment): per-element guarded loads, k(m)-major LDS tiles read back with ds_read_b32.
  // Only shapes whose rows are not 16-byte aligned take it; everything else uses gemm_tn_kernel_t below.
  constexpr int NT = BNT / 64;
  constexpr int LDA_S = BM + 4, LDB_S = BNT + 4;
  constexpr int A_TILE = BK * LDA_S, B_TILE = BK * LDB_S;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * A_TILE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, kk = lane >> 5;
  const DropCfg dc = {g.drop_thr, g.drop_seed, g.drop_scale};
  const int i0 = blockIdx.x * BM;      // along ka
  const int j0 = blockIdx.y * BNT;     // along nb
  const int64_t mbeg = (int64_t)blockIdx.z * g.rows_per_split;
  int64_t mend = mbeg + g.rows_per_split;
  if (mend > g.m) mend = g.m;

  constexpr int AQ = (BK * BM / 4) / 256;    // 4
  constexpr int BQ = (BK * BNT / 4) / 256;   // 4 or 2
  float4 a_reg[AQ], b_reg[BQ];

  auto load_tiles = [&](int64_t mt) {
#pragma unroll
    for (int q = 0; q < AQ; ++q) {
      const int f = tid + 256 * q;
      const int r = f / (BM / 4), c4 = (f % (BM / 4)) * 4;
      const int64_t mrow = mt + r;
      const int ig = i0 + c4;
      a_reg[q] = (mrow < mend) ? load4_guard(g.a + mrow * g.lda + ig, g.ka - ig, g.a_vec) : zero4();
    }
#pragma unroll
    for (int q = 0; q < BQ; ++q) {
      const int f = tid + 256 * q;
      const int r = f / (BNT / 4), c4 = (f % (BNT / 4)) * 4;
      const int64_t mrow = mt + r;
      const int jg = j0 + c4;
      float4 v = zero4();
      if (mrow < mend) {
        const int64_t src = g.b_rows ? g.b_rows[mrow] : mrow;
        v = load4_guard(g.b + src * g.ldb + jg, g.nb - jg, g.b_vec);
        if (g.b_scale) v = xform4(v, g.b_scale, g.b_shift, jg, g.nb, mrow, dc);
      }
      b_reg[q] = v;
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int q = 0; q < AQ; ++q) {
      const int f = tid + 256 * q;
      *reinterpret_cast<float4*>(As + buf * A_TILE + (f / (BM / 4)) * LDA_S + (f % (BM / 4)) * 4) = a_reg[q];
    }
#pragma unroll
    for (int q = 0; q < BQ; ++q) {
      const int f = tid + 256 * q;
      *reinterpret_cast<float4*>(Bs + buf * B_TILE + (f / (BNT / 4)) * LDB_S + (f % (BNT / 4)) * 4) = b_reg[q];
    }
  };

  f32x16 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int64_t nkt = (mend - mbeg + BK - 1) / BK;
  if (nkt > 0) {
    load_tiles(mbeg);
    store_tiles(0);
  }
  __syncthreads();
  for (int64_t kt = 0; kt < nkt; ++kt) {
    const int cur = (int)(kt & 1);
    if (kt + 1 < nkt) load_tiles(mbeg + (kt + 1) * BK);
    const float* as = As + cur * A_TILE + kk * LDA_S + wm * 64 + li;
    const float* bs = Bs + cur * B_TILE + kk * LDB_S + wn * (BNT / 2) + li;
#pragma unroll
    for (int s = 0; s < BK / 2; ++s) {
      float af[2], bf[NT];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = as[(2 * s) * LDA_S + i * 32];
#pragma unroll
      for (int j = 0; j < NT; ++j) bf[j] = bs[(2 * s) * LDB_S + j * 32];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nkt) store_tiles(cur ^ 1);
    __syncthreads();
  }

  float* cbase = g.c + (g.splits > 1 ? (int64_t)blockIdx.z * g.ka * g.ldc : 0);
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = j0 + wn * (BNT / 2) + j * 32 + li;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        if (col < g.nb && row < g.ka) cbase[(int64_t)row * g.ldc + col] = acc[i][j][r];
      }
  }
}

// split-K finish: c[m,n] = epi( sum_z ws[z][m][n] ), fixed order => deterministic
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const GemmArgs g) {
  const int64_t total = g.m * g.n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / g.n;
    const int col = (int)(i - row * g.n);
    float v = 0.f;
    const float* __restrict__ wsz = g.ws;
#pragma unroll 8
    for (int z = 0; z < g.ksplits; ++z) v += wsz[(int64_t)z * total + i];
    if (g.row_scale) v *= g.row_scale[row];
    v = fmaf(v, g.ep_scale ? g.ep_scale[col] : 1.f, g.ep_shift ? g.ep_shift[col] : 0.f);
    if (g.relu) v = fmaxf(v, 0.f);
    g.c[row * g.ldc + col] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// TN, transposing loader (the fast weight-gradient path): C[i,j] = sum_m A[m,i] * B'[m,j].
// Both operands are row-major along the NON-reduction index in global memory, which is the wrong way round
// for the MFMA fragments (a lane wants 4 consecutive reduction terms).  Instead of reading them back with
// 4x as many ds_read_b32 (the first TN kernel: 60 % MFMA busy), every thread loads a 4(m) x 4(col) block,
// transposes it in registers and stores four 16-byte rows into [col][m] LDS tiles -- after that the main loop
// is EXACTLY the NT kernel's (conflict-free ds_read_b128 fragments, 4 MFMAs per read).
// Thread map: tid -> (rg: rows 4rg..4rg+3 of the 32-row k-tile, cg: columns 4cg..4cg+3), interleaved (below).
// ---------------------------------------------------------------------------------------------
template <int BMT, int BNT, int XF, bool ROWS>
__device__ __forceinline__ void gemm_tn_tile_t(const GemmTnArgs& g, const int bx, const int by, const int bz) {
  constexpr int MI = BMT / 64;
  constexpr int NT = BNT / 64;
  constexpr int A_TILE = BMT * LDS_K, B_TILE = BNT * LDS_K;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                      // [2][BM][LDS_K]   rows = output row index i, k = m
  float* Bs = smem + 2 * A_TILE;         // [2][BNT][LDS_K]  rows = output col index j

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, kk = lane >> 5;
  const int i0 = bx * BMT, j0 = by * BNT;
  const int64_t mbeg = (int64_t)bz * g.rows_per_split;
  int64_t mend = mbeg + g.rows_per_split;
  if (mend > g.m) mend = g.m;

  // interleaved map: within every 8 consecutive lanes rg takes 4 values and cg 2, so the four transposed b128
  // stores of a lane group land on 8 distinct 16-byte slots (brute-forced: conflict-free; the plain
  // tid/32, tid%32 map is 4-way conflicted); a wave still reads 256 contiguous bytes per global row
  const int rg = (tid & 3) | (((tid >> 7) & 1) << 2), cg = (tid >> 2) & 31;
  const bool a_active = cg < BMT / 4;                  // 64-wide tiles: half the column groups
  const bool b_active = cg < BNT / 4;
  int a_col = i0 + 4 * cg, b_col = j0 + 4 * cg;
  const int kap = (g.ka + 3) & ~3, nbp = (g.nb + 3) & ~3;
  if (a_col > kap - 4) a_col = kap - 4;               // clamped columns only feed outputs that are never stored
  if (b_col > nbp - 4) b_col = nbp - 4;
  float4 sc4 = zero4(), sh4 = zero4();
  if (XF && b_active) {
    sc4 = ld4g(g.b_scale + b_col);
    sh4 = ld4g(g.b_shift + b_col);
  }
  float4 a_reg[4], b_reg[4];
  int64_t mt_cur = 0;
  int64_t src_row[4];        // ROWS: gathered source rows of the NEXT B loads, fetched one tile ahead (no dependent load in the loop)
  auto load_rows = [&](int64_t mt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int64_t mrow = mt + 4 * rg + r;
      if (mrow > g.m - 1) mrow = g.m - 1;
      src_row[r] = g.b_rows[mrow];
    }
  };

  // Staging in pieces, so that the main loop can put them between MFMAs: 8 global loads (4 reduction rows x {A, B}), the
  // per-row operand transform / split-end mask, and 4 x {A, B} transposed b128 stores (store c = column c of the 4x4 block,
  // so it needs all four rows).
  auto load_piece = [&](int p) {
    const int r = p & 3;
    int64_t mrow = mt_cur + 4 * rg + r;
    if (mrow > g.m - 1) mrow = g.m - 1;
    if (p < 4) {
      if (a_active) a_reg[r] = ld4g(g.a + mrow * g.lda + a_col);
    } else if (b_active) {
      if (ROWS) {
        b_reg[r] = ld4g(g.b + src_row[r] * g.ldb + b_col);
        int64_t nrow = mrow + BK;                       // the row this lane loads one tile later
        if (nrow > g.m - 1) nrow = g.m - 1;
        src_row[r] = g.b_rows[nrow];
      } else {
        b_reg[r] = ld4g(g.b + mrow * g.ldb + b_col);
      }
    }
  };
  auto load_tiles = [&](int64_t mt) {
    mt_cur = mt;
#pragma unroll
    for (int p = 0; p < 8; ++p) load_piece(p);
  };
  float av[4][4], bv[4][4];
  auto prep_row = [&](int r) {
    const bool in = mt_cur + 4 * rg + r < mend;        // rows past this split's end are reduction terms: zero them
    float4 a = in ? a_reg[r] : zero4();
    av[r][0] = a.x; av[r][1] = a.y; av[r][2] = a.z; av[r][3] = a.w;
    if (b_active) {
      float4 b = b_reg[r];
      if (XF) {
        b.x = fmaxf(fmaf(b.x, sc4.x, sh4.x), 0.f);
        b.y = fmaxf(fmaf(b.y, sc4.y, sh4.y), 0.f);
        b.z = fmaxf(fmaf(b.z, sc4.z, sh4.z), 0.f);
        b.w = fmaxf(fmaf(b.w, sc4.w, sh4.w), 0.f);
        if (XF == 2) {
          const uint32_t row = (uint32_t)(mt_cur + 4 * rg + r), c = (uint32_t)(j0 + 4 * cg);
          b.x = glnn::drop_keep(g.drop_seed, g.drop_thr, row, c + 0) ? b.x * g.drop_scale : 0.f;
          b.y = glnn::drop_keep(g.drop_seed, g.drop_thr, row, c + 1) ? b.y * g.drop_scale : 0.f;
          b.z = glnn::drop_keep(g.drop_seed, g.drop_thr, row, c + 2) ? b.z * g.drop_scale : 0.f;
          b.w = glnn::drop_keep(g.drop_seed, g.drop_thr, row, c + 3) ? b.w * g.drop_scale : 0.f;
        }
      }
      if (!in) b = zero4();
      bv[r][0] = b.x; bv[r][1] = b.y; bv[r][2] = b.z; bv[r][3] = b.w;
    }
  };
  auto store_col = [&](int buf, int c) {
    float* as = As + buf * A_TILE + (4 * cg) * LDS_K + 4 * rg;
    float* bs = Bs + buf * B_TILE + (4 * cg) * LDS_K + 4 * rg;
    if (a_active) *reinterpret_cast<float4*>(as + c * LDS_K) = make_float4(av[0][c], av[1][c], av[2][c], av[3][c]);
    if (b_active) *reinterpret_cast<float4*>(bs + c * LDS_K) = make_float4(bv[0][c], bv[1][c], bv[2][c], bv[3][c]);
  };

  f32x16 acc[MI][NT];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  constexpr int G = BK / 8;                 // k-groups per tile (k = 8 kg + 4 kk + t)
  constexpr int MG = 4 * MI * NT;           // MFMAs per k-group
  float fa[2][MI][4], fb[2][NT][4];
  auto read_frags = [&](int buf, int kg, float (&af)[MI][4], float (&bf)[NT][4]) {
    const float* as = As + buf * A_TILE + (wm * (BMT / 2) + li) * LDS_K + kk * 4;
    const float* bs = Bs + buf * B_TILE + (wn * (BNT / 2) + li) * LDS_K + kk * 4;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const float4 v = *reinterpret_cast<const float4*>(as + i * 32 * LDS_K + kg * 8);
      af[i][0] = v.x; af[i][1] = v.y; af[i][2] = v.z; af[i][3] = v.w;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float4 v = *reinterpret_cast<const float4*>(bs + j * 32 * LDS_K + kg * 8);
      bf[j][0] = v.x; bf[j][1] = v.y; bf[j][2] = v.z; bf[j][3] = v.w;
    }
  };
  auto mfma_range = [&](const float (&af)[MI][4], const float (&bf)[NT][4], int m_beg, int m_end) {
#pragma unroll
    for (int mm = m_beg; mm < m_end; ++mm) {
      const int t = mm / (MI * NT), i = (mm / NT) % MI, j = mm % NT;
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][t], bf[j][t], acc[i][j], 0, 0, 0);
    }
  };

  // Same software pipeline as gemm_kernel_fast (see there): fragment reads one k-group ahead; tile t+1 goes registers ->
  // LDS between the MFMAs of group G-2, the loads of tile t+2 are issued between those of the first half of group G-1,
  // the barrier sits inside group G-1 and the next tile's first fragment reads hide behind its second half.  Rows past the
  // split's end are zeroed by prep_row and re-read clamped, in-bounds addresses: unconditional, one basic block.
  const int64_t nkt = (mend - mbeg + BK - 1) / BK;
  if (ROWS && b_active) load_rows(mbeg);
  load_tiles(mbeg);
#pragma unroll
  for (int r = 0; r < 4; ++r) prep_row(r);
#pragma unroll
  for (int c = 0; c < 4; ++c) store_col(0, c);
  __syncthreads();
  read_frags(0, 0, fa[0], fb[0]);
  load_tiles(mbeg + BK);
  for (int64_t kt = 0; kt < nkt; ++kt) {
    const int cur = (int)(kt & 1);
#pragma unroll
    for (int kg = 0; kg + 2 < G; ++kg) {
      read_frags(cur, kg + 1, fa[(kg + 1) & 1], fb[(kg + 1) & 1]);
      mfma_range(fa[kg & 1], fb[kg & 1], 0, MG);
    }
    read_frags(cur, G - 1, fa[(G - 1) & 1], fb[(G - 1) & 1]);
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      __builtin_amdgcn_sched_barrier(0);
      if (p < 4) prep_row(p); else store_col(cur ^ 1, p - 4);
      __builtin_amdgcn_sched_barrier(0);
      mfma_range(fa[(G - 2) & 1], fb[(G - 2) & 1], p * MG / 8, (p + 1) * MG / 8);
    }
    mt_cur = mbeg + (kt + 2) * BK;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      __builtin_amdgcn_sched_barrier(0);
      load_piece(p);
      __builtin_amdgcn_sched_barrier(0);
      mfma_range(fa[(G - 1) & 1], fb[(G - 1) & 1], p * (MG / 2) / 8, (p + 1) * (MG / 2) / 8);
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    read_frags(cur ^ 1, 0, fa[0], fb[0]);
    __builtin_amdgcn_sched_barrier(0);
    mfma_range(fa[(G - 1) & 1], fb[(G - 1) & 1], MG / 2, MG);
  }

  float* cbase = g.c + (g.splits > 1 ? (int64_t)bz * g.ka * g.ldc : 0);
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = j0 + wn * (BNT / 2) + j * 32 + li;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + wm * (BMT / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        if (col < g.nb && row < g.ka) cbase[(int64_t)row * g.ldc + col] = acc[i][j][r];
      }
  }
}

template <int BMT, int BNT, int XF, bool ROWS>
__global__ __launch_bounds__(256) void gemm_tn_kernel_t(const GemmTnArgs g) {
  gemm_tn_tile_t<BMT, BNT, XF, ROWS>(g, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

// ONE launch for several independent weight gradients (the deferred dW_l of a small-batch student step, mlp_step.hip): the blocks of
// every problem's (gi, gj, splits) grid are laid end to end; a block finds its problem and runs the SAME 64 x 64 tile code with that
// problem's arguments -- bit-identical partials, three launches and three dependent-latency chains fewer per step.
constexpr int kTnMultiMax = 4;
struct TnMultiArgs {
  GemmTnArgs g[kTnMultiMax];
  int start[kTnMultiMax + 1];             // first block of every problem
  int gi[kTnMultiMax], gj[kTnMultiMax];
  int xf[kTnMultiMax], rows[kTnMultiMax];
  int n;
};
__global__ __launch_bounds__(256) void gemm_tn_multi_kernel(const TnMultiArgs mm) {
  int p = 0;
#pragma unroll
  for (int q = 1; q < kTnMultiMax; ++q)
    if (q < mm.n && (int)blockIdx.x >= mm.start[q]) p = q;
  const int local = (int)blockIdx.x - mm.start[p];
  const int bx = local % mm.gi[p], by = (local / mm.gi[p]) % mm.gj[p], bz = local / (mm.gi[p] * mm.gj[p]);
  const GemmTnArgs& g = mm.g[p];
  const int xf = mm.xf[p];
  if (mm.rows[p]) {
    if (xf == 0) gemm_tn_tile_t<64, 64, 0, true>(g, bx, by, bz);
    else if (xf == 1) gemm_tn_tile_t<64, 64, 1, true>(g, bx, by, bz);
    else gemm_tn_tile_t<64, 64, 2, true>(g, bx, by, bz);
  } else {
    if (xf == 0) gemm_tn_tile_t<64, 64, 0, false>(g, bx, by, bz);
    else if (xf == 1) gemm_tn_tile_t<64, 64, 1, false>(g, bx, by, bz);
    else gemm_tn_tile_t<64, 64, 2, false>(g, bx, by, bz);
  }
}

// PIPE form of the 128 x 128 weight-gradient tile (see pipe_mainloop): both operands are k(m)-major, so both take the
// KROW path -- no register transpose, no VALU in the loop.  Plain B, no gather, every split a whole number of k-tiles.
template <bool AX>
__device__ __forceinline__ void gemm_tn_pipe_body(const GemmTnArgs& g) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, kk = lane >> 5;
  const int i0 = blockIdx.x * 128, j0 = blockIdx.y * 128;
  const int64_t mbeg = (int64_t)blockIdx.z * g.rows_per_split;
  int64_t mend = mbeg + g.rows_per_split;
  if (mend > g.m) mend = g.m;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if constexpr (AX) {
    const bool own = blockIdx.y == 0 && blockIdx.z == 0;      // the workgroups that also store dgamma / dbeta (/ the zero bias gradient) of their columns
    const PipeAx ax = g.ax_alpha ? PipeAx{g.ax_z + mbeg * g.ax_ldz + i0, g.ax_ldz, g.ax_alpha + i0, g.ax_beta + i0, g.ax_gamma + i0,
                                          nullptr, nullptr, 0, 0, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}
                                 : PipeAx{g.ax_z + mbeg * g.ax_ldz + i0, g.ax_ldz, nullptr, nullptr, nullptr, g.ax_p1 + i0, g.ax_p2 + i0, g.ax_nparts,
                                          (int64_t)g.ka, g.ax_inv_rows, g.ax_bn_gamma + i0, g.ax_bn_mean + i0, g.ax_bn_rstd + i0,
                                          own ? g.ax_dgamma + i0 : nullptr, own ? g.ax_dbeta + i0 : nullptr,
                                          (own && g.ax_colsum) ? g.ax_colsum + i0 : nullptr};
    pipe_mainloop<KROW, KROW, true>(g.a + mbeg * g.lda + i0, g.lda, ((g.ka + 3) & ~3) - i0, g.b + mbeg * g.ldb + j0, g.ldb,
                                    ((g.nb + 3) & ~3) - j0, mend - mbeg, (int)((mend - mbeg + BK - 1) / BK), acc, &ax);
  } else {
    pipe_mainloop<KROW, KROW>(g.a + mbeg * g.lda + i0, g.lda, ((g.ka + 3) & ~3) - i0, g.b + mbeg * g.ldb + j0, g.ldb,
                              ((g.nb + 3) & ~3) - j0, mend - mbeg, (int)((mend - mbeg + BK - 1) / BK), acc);
  }
  float* cbase = g.c + (g.splits > 1 ? (int64_t)blockIdx.z * g.ka * g.ldc : 0);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = j0 + wn * 64 + j * 32 + li;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        if (col < g.nb && row < g.ka) cbase[(int64_t)row * g.ldc + col] = acc[i][j][r];
      }
  }
}

__global__ __launch_bounds__(256) void gemm_tn_kernel_pipe(const GemmTnArgs g) { gemm_tn_pipe_body<false>(g); }
// the same product with A = alpha * a + beta * z + gamma evaluated on the staged pieces (glnn::BnApplyA; pipe_mainloop<.., AX>)
__global__ __launch_bounds__(256) void gemm_tn_kernel_pipe_ax(const GemmTnArgs g) { gemm_tn_pipe_body<true>(g); }

// sum the split partials: c[i] = sum_s ws[s][i]   (fixed order => deterministic)
// Many splits (the weight gradient over tens of thousands of rows: 64 slabs of a 256 x 256 output): one thread per float4 walking
// all slabs keeps a quarter of the CUs busy with 64 dependent-looking loads each (16.8 MB in 16.6 us).  Here a workgroup owns 64
// float4 columns x 4 slab lanes; lane j sums the slabs j, j+4, ... (fixed order), the four lane sums are folded in fixed order
// through LDS: all CUs busy, 16 loads per thread.
__global__ __launch_bounds__(256) void split_reduce_lanes_kernel(const float* __restrict__ ws, int64_t slab, int splits,
                                                                 float* __restrict__ c, int64_t ldc, int ka, int nb) {
  const int64_t total4 = ((int64_t)ka * nb) >> 2;
  const int nb4 = nb >> 2;
  const int lc = threadIdx.x & 63, kl = threadIdx.x >> 6;
  __shared__ float4 sh[4][64];
  for (int64_t i0 = (int64_t)blockIdx.x * 64; i0 < total4; i0 += (int64_t)gridDim.x * 64) {
    const int64_t i = i0 + lc;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < total4) {
#pragma unroll 4
      for (int k = kl; k < splits; k += 4) {
        const float4 v = *reinterpret_cast<const float4*>(ws + k * slab + 4 * i);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
    }
    sh[kl][lc] = s;
    __syncthreads();
    if (kl == 0 && i < total4) {
      const float4 a = sh[0][lc], b = sh[1][lc], d = sh[2][lc], e = sh[3][lc];
      const int64_t r = i / nb4, cc = (i - r * nb4) * 4;
      *reinterpret_cast<float4*>(c + r * ldc + cc) =
          make_float4((a.x + b.x) + (d.x + e.x), (a.y + b.y) + (d.y + e.y), (a.z + b.z) + (d.z + e.z), (a.w + b.w) + (d.w + e.w));
    }
    __syncthreads();
  }
}

__global__ void split_reduce_kernel(const float* __restrict__ ws, int64_t slab, int splits, float* __restrict__ c,
                                    int64_t ldc, int ka, int nb) {
  const int64_t total = (int64_t)ka * nb;
  if (((nb | ldc | slab) & 3) == 0 && glnn::aligned16(ws) && glnn::aligned16(c)) {       // float4 path (workgroup-uniform)
    const int64_t total4 = total >> 2;
    const int nb4 = nb >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
      float4 s = *reinterpret_cast<const float4*>(ws + 4 * i);
#pragma unroll 8
      for (int k = 1; k < splits; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(ws + k * slab + 4 * i);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      const int64_t r = i / nb4, cc = (i - r * nb4) * 4;
      *reinterpret_cast<float4*>(c + r * ldc + cc) = s;
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
#pragma unroll 8
    for (int k = 0; k < splits; ++k) s += ws[k * slab + i];
    const int64_t r = i / nb, cc = i - r * nb;
    c[r * ldc + cc] = s;
  }
}

// column sums of B [m, nb]: stage 1 -> partial[blk][nb], stage 2 -> out[nb]  (deterministic)
__global__ __launch_bounds__(256) void colsum_stage1(const float* __restrict__ b, int64_t ldb, int64_t m, int nb,
                                                      int64_t rows_per_blk, float* __restrict__ partial) {
  const int lc = threadIdx.x & 63;
  const int col = blockIdx.x * 64 + lc;
  const int colc = col < nb ? col : nb - 1;
  const int rlane = threadIdx.x >> 6;  // 0..3
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_blk;
  int64_t r1 = r0 + rows_per_blk;
  if (r1 > m) r1 = m;
  float s = 0.f;
  // 8 independent (clamped, always valid) loads in flight per step instead of one dependent load per iteration
  for (int64_t rb = r0 + rlane; rb < r1; rb += 32) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t r = rb + 4 * u;
      v[u] = b[(r < r1 ? r : r0) * ldb + colc];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (rb + 4 * u < r1) ? v[u] : 0.f;
  }
  __shared__ float sh[4][64];
  sh[rlane][lc] = s;
  __syncthreads();
  if (rlane == 0 && col < nb && r0 < m) partial[(int64_t)blockIdx.y * nb + col] = (sh[0][lc] + sh[1][lc]) + (sh[2][lc] + sh[3][lc]);
}
__global__ void colsum_stage2(const float* __restrict__ partial, int nblk, int nb, float* __restrict__ out) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= nb) return;
  float s = 0.f;
#pragma unroll 8
  for (int k = 0; k < nblk; ++k) s += partial[(int64_t)k * nb + col];
  out[col] = s;
}

template <typename K>
int set_smem(K kernel, size_t bytes) {
  if (bytes > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
      return glnn::fail(GLNN_ERR_HIP, "hipFuncSetAttribute(max dynamic LDS=%zu) failed", bytes);
  }
  return GLNN_OK;
}

template <typename K>
int launch_gemm_kernel(K kernel, int& configured, size_t smem, const GemmArgs& g, int bn, hipStream_t st, int bm = BM) {
  if (configured > 0) configured = set_smem(kernel, smem);
  if (configured != GLNN_OK) return configured;
  const int64_t gm = (g.m + bm - 1) / bm;
  const int gn = (g.n + bn - 1) / bn;
  if (gm > 0x7fffffffLL) return glnn::fail(GLNN_ERR_UNSUPPORTED, "glnn_gemm_f32: m too large");
  hipLaunchKernelGGL(kernel, dim3((unsigned)gm, (unsigned)gn, (unsigned)g.ksplits), dim3(256), smem, st, g);
  int rc = glnn::check_launch("glnn_gemm_f32");
  if (rc == GLNN_OK && g.ksplits > 1 && !g.defer_fold) {
    int64_t blocks = (g.m * g.n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)blocks), dim3(256), 0, st, g);
    rc = glnn::check_launch("glnn_gemm_f32(split-k epilogue)");
  }
  return rc;
}

#ifndef GLNN_GEMM_BKF
#define GLNN_GEMM_BKF 32
#endif
#ifndef GLNN_GEMM_BMT
#define GLNN_GEMM_BMT BM
#endif
template <int BN, bool B_KN>
int launch_gemm(GemmArgs& g, bool fast, hipStream_t st) {
  constexpr int BKF = GLNN_GEMM_BKF;
  constexpr int BMT = GLNN_GEMM_BMT;
  constexpr size_t smem_generic = sizeof(float) * 2 * (BM * LDS_K + (B_KN ? BK * (BN + 4) : BN * LDS_K));
  constexpr size_t smem_fast = sizeof(float) * 2 * (BMT * (BKF + 4) + (B_KN ? BKF * (BN + 4) : BN * (BKF + 4)));
  static int cfg[4] = {1, 1, 1, 1};   // >0 = not configured yet
  if (!fast) {
    g.ksplits = 1; g.ktiles_per_split = (g.k + BK - 1) / BK;
    return launch_gemm_kernel(gemm_kernel_generic<BN, B_KN>, cfg[3], smem_generic, g, BN, st);
  }
  if constexpr (BN == 128 && BMT == 128) {
    // plain operands, whole k-tiles, no gather, and every byte offset inside the 2 GiB descriptor window
    const bool window = B_KN ? (g.lda < (1 << 21) && (int64_t)g.k * g.ldb < (1 << 28)) : (g.lda < (1 << 21) && g.ldb < (1 << 21));
    if (pipe_enabled() && !g.a_scale && g.ksplits == 1 && g.k % BK == 0 && !g.a_rows && window) {
      static int cfg_pipe = 1;
      if (g.st_done) *g.st_done = 1;
      return launch_gemm_kernel(gemm_kernel_pipe<B_KN>, cfg_pipe, pipe_lds_bytes(ROWK, B_KN ? KROW : ROWK), g, BN, st, 128);
    }
  }
  g.st_mean = g.st_m2 = nullptr;
  if (BKF != BK) g.ktiles_per_split *= BK / BKF;     // split bookkeeping is in units of the fast kernel's k-tiles
  if (!g.a_scale) return launch_gemm_kernel(gemm_kernel_fast<BMT, BN, B_KN, 0, BKF>, cfg[0], smem_fast, g, BN, st, BMT);
  if (!g.drop_thr) return launch_gemm_kernel(gemm_kernel_fast<BMT, BN, B_KN, 1, BKF>, cfg[1], smem_fast, g, BN, st, BMT);
  return launch_gemm_kernel(gemm_kernel_fast<BMT, BN, B_KN, 2, BKF>, cfg[2], smem_fast, g, BN, st, BMT);
}

// 64 x 64 tiles (one MFMA block per wave): the latency form for outputs of a few dozen tiles
#ifndef GLNN_GEMM_SMALL_BKF
#define GLNN_GEMM_SMALL_BKF 32
#endif
template <bool B_KN>
int launch_gemm_small(GemmArgs& g, hipStream_t st) {
  constexpr int BKF = GLNN_GEMM_SMALL_BKF;
  constexpr size_t smem = sizeof(float) * 2 * (64 * (BKF + 4) + (B_KN ? BKF * (64 + 4) : 64 * (BKF + 4)));
  static int cfg[3] = {1, 1, 1};
  if (BKF < BK) g.ktiles_per_split *= BK / BKF;
  if (BKF > BK) g.ktiles_per_split = (g.ktiles_per_split + BKF / BK - 1) / (BKF / BK);
  if (!g.a_scale) return launch_gemm_kernel(gemm_kernel_fast<64, 64, B_KN, 0, BKF>, cfg[0], smem, g, 64, st, 64);
  if (!g.drop_thr) return launch_gemm_kernel(gemm_kernel_fast<64, 64, B_KN, 1, BKF>, cfg[1], smem, g, 64, st, 64);
  return launch_gemm_kernel(gemm_kernel_fast<64, 64, B_KN, 2, BKF>, cfg[2], smem, g, 64, st, 64);
}

}  // namespace

static int gemm_impl(const float* a, int64_t lda, const int64_t* a_rows, const float* a_scale,
                     const float* a_shift, float drop_p, uint32_t drop_seed, int64_t m, int k, const float* b, int64_t ldb, int b_layout, int n,
                     const float* row_scale, const float* ep_scale, const float* ep_shift, int relu, float* c,
                     int64_t ldc, float* workspace, int64_t workspace_floats, void* stream, int* defer_splits, glnn::ColStats* cs = nullptr) {
  GLNN_REQUIRE(a && b && c, "glnn_gemm_f32: null pointer");
  GLNN_REQUIRE(m >= 0 && k >= 1 && n >= 1, "glnn_gemm_f32: bad sizes m=%lld k=%d n=%d", (long long)m, k, n);
  GLNN_REQUIRE(lda >= k && ldc >= n, "glnn_gemm_f32: lda/ldc too small");
  GLNN_REQUIRE(b_layout == 0 || b_layout == 1, "glnn_gemm_f32: b_layout must be 0 ([n,k]) or 1 ([k,n])");
  GLNN_REQUIRE(ldb >= (b_layout ? n : k), "glnn_gemm_f32: ldb too small");
  GLNN_REQUIRE((a_scale == nullptr) == (a_shift == nullptr), "glnn_gemm_f32: a_scale and a_shift go together");
  GLNN_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || a_scale), "glnn_gemm_f32: drop_p in [0,1) and needs a_scale/a_shift");
  if (m == 0) return GLNN_OK;
  GemmArgs g;
  g.drop_thr = glnn::drop_threshold(drop_p); g.drop_seed = drop_seed; g.drop_scale = 1.0f / (1.0f - drop_p);
  g.a = a; g.lda = lda; g.a_rows = a_rows; g.a_scale = a_scale; g.a_shift = a_shift; g.m = m; g.k = k;
  g.b = b; g.ldb = ldb; g.n = n; g.row_scale = row_scale; g.ep_scale = ep_scale; g.ep_shift = ep_shift; g.relu = relu;
  g.c = c; g.ldc = ldc;
  // float4 loads are legal when every row start is 16-byte aligned; the tail (k or n not a multiple of 4)
  // is handled per element inside load4_guard, which needs the padded part of the row to be readable:
  // true for A when lda >= roundup4(k); for B only when ldb >= roundup4(extent).
  g.a_vec = (lda % 4 == 0) && glnn::aligned16(a);
  g.b_vec = (ldb % 4 == 0) && glnn::aligned16(b);
  g.ksplits = 1; g.ktiles_per_split = (k + BK - 1) / BK; g.ws = nullptr; g.defer_fold = 0;
  g.st_mean = g.st_m2 = nullptr; g.st_done = nullptr;
  g.bd_z = nullptr; g.bd_ldz = 0; g.bd_scale = g.bd_shift = g.bd_mean = g.bd_rstd = nullptr; g.bd_thr = g.bd_seed = 0u; g.bd_dscale = 1.f;
  g.bd_relu = 0; g.bd_s1 = g.bd_s2 = nullptr;
  int tile_stats_done = 0;
  if (cs && !relu && !row_scale && !defer_splits && cs->ws && cs->ws_floats >= 2 * ((m + BM - 1) / BM) * (int64_t)n) {
    g.st_mean = cs->ws; g.st_m2 = cs->ws + ((m + BM - 1) / BM) * (int64_t)n; g.st_done = &tile_stats_done;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // fast path: all float4 loads legal and all-or-nothing at the k (and, for [k,n], n) boundary
  const bool fast = g.a_vec && g.b_vec && lda >= ((k + 3) & ~3) && ldb >= (((b_layout ? n : k) + 3) & ~3) &&
                    (!a_scale || (k % 4 == 0 && glnn::aligned16(a_scale) && glnn::aligned16(a_shift)));
  // W[n, k] with rows off the 16-byte grid (a feature width that is not a multiple of 4: cora's 1433-wide first layers) and a plain
  // epilogue: the dword-loading latency kernel of mlp_lat.hip instead of the guarded generic one (2485 x 64 x 1433: 240 us)
  // W[n, k] whose rows are not float4-addressable over a deep reduction (cora 1433, citeseer 3703, penn94 4814 features), a caller with
  // workspace to spare: a padded shadow in the workspace's tail (one small launch) and the tiled kernels, split-K included -- the
  // unaligned-W latency kernel below walks all of K in (m / 32) x (n / 32) workgroups (citeseer's SAGE projection: 107 us)
  if (!fast && !defer_splits && !b_layout && g.a_vec && lda >= ((k + 3) & ~3) && !g.b_vec && k >= 512 && m >= 256 && workspace &&
      glnn::aligned16(workspace)) {
    const int64_t kp = (k + 3) & ~(int64_t)3, need = ((int64_t)n * kp + 3) & ~(int64_t)3;
    const int64_t left = (workspace_floats - need) & ~(int64_t)3;
    if (left >= 2 * m * n) {
      float* shadow = workspace + left;
      const int rc = glnn::pad_rows(b, ldb, n, k, shadow, kp, stream);
      if (rc != GLNN_OK) return rc;
      return gemm_impl(a, lda, a_rows, a_scale, a_shift, drop_p, drop_seed, m, k, shadow, kp, 0, n, row_scale, ep_scale, ep_shift, relu, c, ldc,
                       workspace, left, stream, defer_splits);
    }
  }
  if (!fast && !defer_splits && !b_layout && g.a_vec && lda >= ((k + 3) & ~3) && !g.b_vec && !a_scale && !row_scale) {
    const int rc = glnn::gemm_lat(a, lda, a_rows, nullptr, nullptr, 0.f, 0u, m, k, b, ldb, 0, n, ep_shift, c, ldc, nullptr, nullptr, nullptr, stream,
                                  nullptr, 0, ep_scale, relu);
    if (rc != GLNN_ERR_UNSUPPORTED) return rc;
  }
  // a short reduction over many rows (k <= 128: a projection of feature / aggregate rows): persistent workgroups with the weight panel
  // resident in LDS (gemm_rowpanel.hip) instead of one workgroup per output tile with 3-4 k-tiles each
  if (fast && !defer_splits && !b_layout && !a_rows && !a_scale && !row_scale && rowpanel_enabled()) {
    const int rc = glnn::gemm_rowpanel(a, lda, m, k, b, ldb, n, ep_scale, ep_shift, relu, c, ldc, stream, cs);
    if (rc != GLNN_ERR_UNSUPPORTED) return rc;
  }
  // latency regime: fewer than 64 tiles of 128 x (128|64) -> 64 x 64 tiles, four times the workgroups, a quarter of the
  // serial MFMA chain per k-tile; split-K only when the chain is still long (>= 16 k-tiles)
  {
    const int bn = n > 64 ? 128 : 64;
    const int64_t tiles = ((m + BM - 1) / BM) * ((n + bn - 1) / bn);
    if (fast && tiles <= 64) {
      const int nk = (k + BK - 1) / BK;
      const int64_t tiles_s = ((m + 63) / 64) * ((n + 63) / 64);
      if (workspace && nk >= 16 && tiles_s < 512) {
        int64_t sp = (512 + tiles_s - 1) / tiles_s;
        if (sp > nk / 4) sp = nk / 4;
        if (sp * m * n > workspace_floats) sp = workspace_floats / (m * n);
        if (sp > 1) {
          g.ktiles_per_split = (nk + (int)sp - 1) / (int)sp;
          g.ksplits = (nk + g.ktiles_per_split - 1) / g.ktiles_per_split;
          g.ws = workspace;
        }
      }
      if (defer_splits) {                       // partials only: the caller's next kernel sums the slabs ws[z][m][n] (and adds the bias)
        if (g.ksplits <= 1) return GLNN_ERR_UNSUPPORTED;
        g.defer_fold = 1;
        *defer_splits = g.ksplits;
      }
      return b_layout ? launch_gemm_small<true>(g, st) : launch_gemm_small<false>(g, st);
    }
  }
  if (defer_splits) return GLNN_ERR_UNSUPPORTED;
  // split-K when the output has too few tiles to fill 256 CUs twice over and K is deep enough
  if (fast && workspace) {
    const int bn = n > 64 ? 128 : 64;
    const int64_t tiles = ((m + BM - 1) / BM) * ((n + bn - 1) / bn);
    const int nk = (k + BK - 1) / BK;
    // a narrow output over a deep reduction and many rows is a STREAM of A (penn94's GCN: 41554 x 4814 -> 64, 0.8 GB): a few hundred
    // workgroups with one k-tile in flight each leave it at 2.3 TB/s; split until ~2048 exist (347 -> 285 us)
    if (n <= 64 && nk >= 64 && tiles >= 256 && tiles < 2048) {
      int64_t sp = (2048 + tiles - 1) / tiles;
      if (sp > nk / 8) sp = nk / 8;
      if (sp * m * n > workspace_floats) sp = workspace_floats / (m * n);
      if (sp > 1) {
        g.ktiles_per_split = (nk + (int)sp - 1) / (int)sp;
        g.ksplits = (nk + g.ktiles_per_split - 1) / g.ktiles_per_split;
        g.ws = workspace;
      }
    } else if (tiles < 256 && nk >= 4) {
      int64_t sp = (512 + tiles - 1) / tiles;
      // big problems keep >= 4 k-tiles per split (the slab round trip must stay small next to the MFMA work); tiny
      // ones (a handful of tiles, e.g. the B=512 student) are pure latency chains of dependent k-tiles: cut them to
      // 2 k-tiles per workgroup
      const int min_tiles_per_split = tiles <= 32 ? 2 : 4;
      if (sp > nk / min_tiles_per_split) sp = nk / min_tiles_per_split;
      if (sp * m * n > workspace_floats) sp = workspace_floats / (m * n);
      if (sp > 1) {
        g.ktiles_per_split = (nk + (int)sp - 1) / (int)sp;
        g.ksplits = (nk + g.ktiles_per_split - 1) / g.ktiles_per_split;
        g.ws = workspace;
      }
    }
  }
  int rc;
  if (n > 64) rc = b_layout ? launch_gemm<128, true>(g, fast, st) : launch_gemm<128, false>(g, fast, st);
  else rc = b_layout ? launch_gemm<64, true>(g, fast, st) : launch_gemm<64, false>(g, fast, st);
  if (rc == GLNN_OK && tile_stats_done) {
    cs->ws_cnt = nullptr; cs->ws_mean = g.st_mean; cs->ws_m2 = g.st_m2;
    cs->nparts = (int)((m + BM - 1) / BM); cs->chunk_rows = BM; cs->done = 1;
  }
  return rc;
}

// glnn_gemm_f32 with plain operands (C = A W^T + bias) that also leaves the first pass of C's column statistics when the kernel taking
// the shape can (the row-panel kernel: per-workgroup triples; the pipelined kernel: per-tile mean / M2) -- see glnn::ColStats
int glnn::gemm_stats(const float* a, int64_t lda, int64_t m, int k, const float* w, int64_t ldw, int n, const float* bias, float* c, int64_t ldc,
                     float* workspace, int64_t workspace_floats, void* stream, glnn::ColStats* cs) {
  if (cs) { cs->done = 0; cs->nparts = 0; cs->chunk_rows = 0; cs->ws_cnt = cs->ws_mean = cs->ws_m2 = nullptr; }
  return gemm_impl(a, lda, nullptr, nullptr, nullptr, 0.f, 0u, m, k, w, ldw, 0, n, nullptr, nullptr, bias, 0, c, ldc, workspace, workspace_floats,
                   stream, nullptr, cs);
}

// dy = masks(a . w) with w [k, n] -- the input gradient of a hidden layer behind its tail's dropout / ReLU masks -- and the per-128-row-tile
// column sums S1 / S2 of the BatchNorm backward (s1 / s2 [ceil(m / 128)][n]) out of the pipelined kernel's epilogue; da itself is never
// written.  GLNN_ERR_UNSUPPORTED (nothing launched) unless the pipelined kernel takes the shape: float4-addressable plain operands,
// k % 32 == 0, n > 64, byte offsets inside the descriptor windows.
int glnn::gemm_bn_dy(const float* a, int64_t lda, int64_t m, int k, const float* w, int64_t ldw, int n, const glnn::BnTail& t, float* c, int64_t ldc,
                     float* s1, float* s2, void* stream) {
  if (!a || !w || !c || !s1 || !s2 || !t.z || !t.mean || !t.rstd || !t.a_scale || !t.a_shift || m < 1 || k < 1 || n <= 64) return GLNN_ERR_UNSUPPORTED;
  if (lda < k || ldw < n || ldc < n || t.ldz < n || t.drop_p < 0.f || t.drop_p >= 1.f) return GLNN_ERR_UNSUPPORTED;
  const bool vec = lda % 4 == 0 && ldw % 4 == 0 && glnn::aligned16(a) && glnn::aligned16(w) && lda >= ((k + 3) & ~3) && ldw >= ((n + 3) & ~3);
  const bool window = lda < (1 << 21) && (int64_t)k * ldw < (1 << 28);
  if (!vec || !window || !pipe_enabled() || k % BK != 0 || (m + 127) / 128 > 0x7fffffffLL) return GLNN_ERR_UNSUPPORTED;
  GemmArgs g = {};
  g.a = a; g.lda = lda; g.m = m; g.k = k; g.b = w; g.ldb = ldw; g.n = n; g.c = c; g.ldc = ldc; g.a_vec = g.b_vec = 1;
  g.ksplits = 1; g.ktiles_per_split = k / BK; g.drop_scale = 1.f;
  g.bd_z = t.z; g.bd_ldz = t.ldz; g.bd_scale = t.a_scale; g.bd_shift = t.a_shift; g.bd_mean = t.mean; g.bd_rstd = t.rstd;
  g.bd_thr = glnn::drop_threshold(t.drop_p); g.bd_seed = t.drop_seed; g.bd_dscale = 1.0f / (1.0f - t.drop_p); g.bd_relu = t.relu ? 1 : 0;
  g.bd_s1 = s1; g.bd_s2 = s2;
  static int cfg_pipe = 1;
  return launch_gemm_kernel(gemm_kernel_pipe<true>, cfg_pipe, pipe_lds_bytes(ROWK, KROW), g, 128, reinterpret_cast<hipStream_t>(stream), 128);
}

extern "C" int glnn_gemm_f32(const float* a, int64_t lda, const int64_t* a_rows, const float* a_scale,
                             const float* a_shift, float drop_p, uint32_t drop_seed, int64_t m, int k, const float* b, int64_t ldb, int b_layout, int n,
                             const float* row_scale, const float* ep_scale, const float* ep_shift, int relu, float* c,
                             int64_t ldc, float* workspace, int64_t workspace_floats, void* stream) {
  return gemm_impl(a, lda, a_rows, a_scale, a_shift, drop_p, drop_seed, m, k, b, ldb, b_layout, n, row_scale, ep_scale, ep_shift, relu, c, ldc,
                   workspace, workspace_floats, stream, nullptr);
}

// The split-K product of the latency regime WITHOUT its fold launch: raw partial sums workspace[z][m][n], z < *splits; the kernel that
// consumes C = sum_z + bias reads the slabs itself (student.hip: the loss kernel).  GLNN_ERR_UNSUPPORTED (nothing launched) when this
// problem would not be split -- the caller then takes glnn_gemm_f32.
int glnn::gemm_split_partials(const float* a, int64_t lda, const int64_t* a_rows, const float* a_scale, const float* a_shift, float drop_p,
                              uint32_t drop_seed, int64_t m, int k, const float* b, int64_t ldb, int b_layout, int n, float* workspace,
                              int64_t workspace_floats, int* splits, void* stream) {
  if (!a || !b || !workspace || !splits || m < 1 || k < 1 || n < 1 || lda < k || ldb < (b_layout ? n : k)) return GLNN_ERR_UNSUPPORTED;
  if ((a_scale == nullptr) != (a_shift == nullptr) || (drop_p > 0.f && !a_scale) || drop_p < 0.f || drop_p >= 1.f) return GLNN_ERR_UNSUPPORTED;
  return gemm_impl(a, lda, a_rows, a_scale, a_shift, drop_p, drop_seed, m, k, b, ldb, b_layout, n, nullptr, nullptr, nullptr, 0, workspace, n,
                   workspace, workspace_floats, stream, splits);
}

// the fold launch of gemm_split_partials for a consumer that cannot read slabs: c = sum_z workspace[z][m][n] + bias
int glnn::gemm_fold_partials(const float* workspace, int splits, int64_t m, int n, const float* bias, float* c, int64_t ldc, void* stream) {
  GLNN_REQUIRE(workspace && c && splits >= 1 && m >= 1 && n >= 1 && ldc >= n, "glnn::gemm_fold_partials: bad arguments");
  GemmArgs g = {};
  g.m = m; g.n = n; g.c = c; g.ldc = ldc; g.ksplits = splits; g.ws = const_cast<float*>(workspace); g.ep_shift = bias;
  int64_t blocks = (m * n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g);
  return glnn::check_launch("glnn::gemm_fold_partials");
}

// ---- several independent weight gradients in ONE gemm launch + ONE fold launch (see gemm_tn_multi_kernel) -------------------
struct FoldMultiArgs {
  const float* ws[kTnMultiMax]; int64_t slab[kTnMultiMax]; int splits[kTnMultiMax];
  float* c[kTnMultiMax]; int64_t ldc[kTnMultiMax]; int ka[kTnMultiMax], nb[kTnMultiMax];
  int start[kTnMultiMax + 1]; int n;
  int lanes[kTnMultiMax];     // 1: the order of split_reduce_lanes_kernel (what glnn_gemm_tn_f32 uses from 16 splits on), else k ascending
};
// split_reduce_kernel for every problem with more than one split: the same sums in the same order (k = 0, 1, 2, ...)
__global__ void split_reduce_multi_kernel(const FoldMultiArgs fm) {
  int p = 0;
#pragma unroll
  for (int q = 1; q < kTnMultiMax; ++q)
    if (q < fm.n && (int)blockIdx.x >= fm.start[q]) p = q;
  const int64_t b0 = (int)blockIdx.x - fm.start[p], nblk = fm.start[p + 1] - fm.start[p];
  const float* ws = fm.ws[p];
  float* c = fm.c[p];
  const int64_t slab = fm.slab[p], ldc = fm.ldc[p];
  const int splits = fm.splits[p], ka = fm.ka[p], nb = fm.nb[p];
  const int64_t total = (int64_t)ka * nb;
  if (((nb | ldc | slab) & 3) == 0 && glnn::aligned16(ws) && glnn::aligned16(c)) {
    const int64_t total4 = total >> 2;
    const int nb4 = nb >> 2;
    for (int64_t i = b0 * blockDim.x + threadIdx.x; i < total4; i += nblk * blockDim.x) {
      float4 s;
      if (fm.lanes[p]) {        // four interleaved lanes k = j, j + 4, ... (each from 0, k ascending), combined (l0 + l1) + (l2 + l3)
        float4 l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) l[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k0 = 0; k0 < splits; k0 += 4) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (k0 + j < splits) {
              const float4 v = *reinterpret_cast<const float4*>(ws + (k0 + j) * slab + 4 * i);
              l[j].x += v.x; l[j].y += v.y; l[j].z += v.z; l[j].w += v.w;
            }
          }
        }
        s = make_float4((l[0].x + l[1].x) + (l[2].x + l[3].x), (l[0].y + l[1].y) + (l[2].y + l[3].y), (l[0].z + l[1].z) + (l[2].z + l[3].z),
                        (l[0].w + l[1].w) + (l[2].w + l[3].w));
      } else {
        s = *reinterpret_cast<const float4*>(ws + 4 * i);
        for (int k = 1; k < splits; ++k) {
          const float4 v = *reinterpret_cast<const float4*>(ws + k * slab + 4 * i);
          s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
      }
      const int64_t r = i / nb4, cc = (i - r * nb4) * 4;
      *reinterpret_cast<float4*>(c + r * ldc + cc) = s;
    }
    return;
  }
  for (int64_t i = b0 * blockDim.x + threadIdx.x; i < total; i += nblk * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += ws[k * slab + i];
    const int64_t r = i / nb, cc = i - r * nb;
    c[r * ldc + cc] = s;
  }
}

// The plan glnn_gemm_tn_f32 makes for a "small" problem (64 x 64 tiles, the reduction split until ~1024 workgroups exist), or false
// if the problem would not take that path there.  `avail` = workspace floats the split slabs may use.
// ONE definition of "the weight gradient takes the 64 x 64 latency tiles", shared by glnn::gemm_tn and the batched plan below (they must
// agree: the batched launch promises the per-problem plan of the single call)
static inline bool tn_small_regime(bool fast, int tiles128, bool pipe_shape, int64_t m, int ka) {
  return fast && tiles128 <= 64 && !(pipe_shape && m >= 2048) && !(m >= 8192 && ka >= 1024);
}

static bool tn_small_plan(int64_t m, int ka, int nb, int64_t lda, int64_t ldb, const float* a, const float* b, const float* b_scale,
                          const float* b_shift, const int64_t* b_rows, int64_t avail, int* gi, int* gj, int* splits, int64_t* rps) {
  const bool a_vec = (lda % 4 == 0) && glnn::aligned16(a), b_vec = (ldb % 4 == 0) && glnn::aligned16(b);
  const bool fast = a_vec && b_vec && lda >= ((ka + 3) & ~3) && ldb >= ((nb + 3) & ~3) &&
                    (!b_scale || (nb % 4 == 0 && glnn::aligned16(b_scale) && glnn::aligned16(b_shift)));
  const int bnt = nb > 64 ? 128 : 64;
  const int gi0 = (ka + BM - 1) / BM, gj0 = (nb + bnt - 1) / bnt;
  const bool pipe_shape = fast && bnt == 128 && !b_scale && !b_rows && pipe_enabled() && (lda > ldb ? lda : ldb) < (1 << 20);
  if (!tn_small_regime(fast, gi0 * gj0, pipe_shape, m, ka)) return false;
  *gi = (ka + 63) / 64; *gj = (nb + 63) / 64;
  int sp = 1;
  const int64_t slab = (int64_t)ka * nb;
  if (*gi * *gj < 1024) {
    sp = (1024 + *gi * *gj - 1) / (*gi * *gj);
    const int64_t max_by_rows = (m + 2 * BK - 1) / (2 * BK);
    if (sp > max_by_rows) sp = (int)max_by_rows;
    if ((int64_t)sp * slab > avail) sp = (int)(avail / slab);
    if (sp < 1) sp = 1;
  }
  int64_t r = (m + sp - 1) / sp;
  r = (r + BK - 1) / BK * BK;
  *rps = r;
  *splits = (int)((m + r - 1) / r);
  return true;
}

namespace glnn {
// n <= 4 independent products c_p = a_p^T . b'_p (the arguments of glnn_gemm_tn_f32, no col_sum_a) as ONE gemm launch and ONE fold
// launch.  Returns GLNN_ERR_UNSUPPORTED without launching anything when a problem would not take the 64 x 64 path of
// glnn_gemm_tn_f32 or the slabs do not fit the workspace: the caller then issues the products one by one.  Results are bit-identical
// to the one-by-one form (same tile code, same split plan computed against the same workspace size, same fold order).
int gemm_tn_batch(const TnProblem* pr, int n, float* workspace, int64_t workspace_floats, void* stream, GradFold* defer) {
  if (n < 1 || n > kTnMultiMax || !workspace) return GLNN_ERR_UNSUPPORTED;
  TnMultiArgs mm;
  FoldMultiArgs fm;
  mm.n = n; fm.n = 0;
  int64_t ws_off = 0;
  int blocks = 0, fblocks = 0;
  for (int p = 0; p < n; ++p) {
    const TnProblem& q = pr[p];
    if (!(q.a && q.b && q.c) || q.m < 1 || q.ka < 1 || q.nb < 1 || q.lda < q.ka || q.ldb < q.nb || q.ldc < q.nb) return GLNN_ERR_UNSUPPORTED;
    if ((q.b_scale == nullptr) != (q.b_shift == nullptr) || q.drop_p < 0.f || q.drop_p >= 1.f || (q.drop_p > 0.f && !q.b_scale)) return GLNN_ERR_UNSUPPORTED;
    int gi, gj, sp; int64_t rps;
    if (!tn_small_plan(q.m, q.ka, q.nb, q.lda, q.ldb, q.a, q.b, q.b_scale, q.b_shift, q.b_rows, workspace_floats, &gi, &gj, &sp, &rps))
      return GLNN_ERR_UNSUPPORTED;
    GemmTnArgs& g = mm.g[p];
    g.drop_thr = glnn::drop_threshold(q.drop_p); g.drop_seed = q.drop_seed; g.drop_scale = 1.0f / (1.0f - q.drop_p);
    g.a = q.a; g.lda = q.lda; g.m = q.m; g.ka = q.ka; g.b = q.b; g.ldb = q.ldb; g.b_rows = q.b_rows; g.b_scale = q.b_scale; g.b_shift = q.b_shift;
    g.nb = q.nb; g.a_vec = 1; g.b_vec = 1; g.splits = sp; g.rows_per_split = rps;
    const int64_t slab = (int64_t)q.ka * q.nb;
    if (sp > 1) {
      ws_off = (ws_off + 3) & ~(int64_t)3;
      if (ws_off + (int64_t)sp * slab > workspace_floats) return GLNN_ERR_UNSUPPORTED;
      g.c = workspace + ws_off; g.ldc = q.nb;
      const int f = fm.n++;
      fm.ws[f] = g.c; fm.slab[f] = slab; fm.splits[f] = sp; fm.c[f] = q.c; fm.ldc[f] = q.ldc; fm.ka[f] = q.ka; fm.nb[f] = q.nb;
      // the fold order glnn_gemm_tn_f32 would use for this problem (its float4 conditions; the slab base is 16-byte aligned by ws_off)
      const bool lanes = sp >= 16 && ((q.nb | q.ldc | slab) & 3) == 0 && glnn::aligned16(g.c) && glnn::aligned16(q.c);
      fm.lanes[f] = lanes ? 1 : 0;
      int fb = (int)((slab + 255) / 256);
      if (fb > 2048) fb = 2048;
      fm.start[f] = fblocks; fblocks += fb; fm.start[f + 1] = fblocks;
      ws_off += (int64_t)sp * slab;
      if (defer) {
        if (q.ldc != q.nb) return GLNN_ERR_UNSUPPORTED;            // the consumer indexes the slabs like the dense tensor
        defer[p] = {q.c, g.c, sp, lanes ? 1 : 0, slab};
      }
    } else {
      g.c = q.c; g.ldc = q.ldc;
      if (defer) defer[p] = {q.c, nullptr, 0, 0, 0};
    }
    mm.start[p] = blocks; mm.gi[p] = gi; mm.gj[p] = gj;
    mm.xf[p] = !q.b_scale ? 0 : (g.drop_thr ? 2 : 1);
    mm.rows[p] = q.b_rows ? 1 : 0;
    blocks += gi * gj * sp;
    mm.start[p + 1] = blocks;
  }
  for (int p = n; p < kTnMultiMax; ++p) { mm.start[p + 1] = blocks; mm.gi[p] = mm.gj[p] = 1; mm.xf[p] = mm.rows[p] = 0; }
  constexpr size_t smem_s = sizeof(float) * 2 * (64 * LDS_K + 64 * LDS_K);
  static int cfg = 1;
  if (cfg > 0) cfg = set_smem(gemm_tn_multi_kernel, smem_s);
  if (cfg != GLNN_OK) return cfg;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(gemm_tn_multi_kernel, dim3(blocks), dim3(256), smem_s, st, mm);
  int rc = glnn::check_launch("glnn_gemm_tn_f32(batch)");
  if (rc != GLNN_OK || fm.n == 0 || defer) return rc;
  for (int f = fm.n; f < kTnMultiMax; ++f) fm.start[f + 1] = fblocks;
  hipLaunchKernelGGL(split_reduce_multi_kernel, dim3(fblocks), dim3(256), 0, st, fm);
  return glnn::check_launch("glnn_gemm_tn_f32(batch fold)");
}
}  // namespace glnn

extern "C" int glnn_gemm_tn_f32(const float* a, int64_t lda, int64_t m, int ka, const float* b, int64_t ldb,
                                const int64_t* b_rows, const float* b_scale, const float* b_shift, float drop_p,
                                uint32_t drop_seed, int nb, float* c, int64_t ldc, float* col_sum_a, float* workspace, int64_t workspace_floats, void* stream) {
  return glnn::gemm_tn(a, lda, m, ka, b, ldb, b_rows, b_scale, b_shift, drop_p, drop_seed, nb, c, ldc, col_sum_a, workspace, workspace_floats,
                       stream, nullptr, nullptr, nullptr, 0);
}

// would gemm_tn(..., bn) take this product (the pipelined 128 x 128 kernel with plain float4-addressable operands)?  The caller decides
// BEFORE it leaves dz unwritten; the one condition not covered is the split of the reduction, which needs rows_per_split x ld < 2^28
// (any workspace that lets the output split into ~256 workgroups does)
bool glnn::gemm_tn_takes_bn(const float* a, int64_t lda, int64_t m, int ka, const float* b, int64_t ldb, int nb, const float* z, int64_t ldz,
                            int64_t workspace_floats) {
  const bool vec = (lda % 4 == 0) && (ldb % 4 == 0) && (ldz % 4 == 0) && glnn::aligned16(a) && glnn::aligned16(b) && glnn::aligned16(z);
  const bool fast = vec && lda >= ((ka + 3) & ~3) && ldb >= ((nb + 3) & ~3) && ldz >= ((ka + 3) & ~3);
  if (!(m >= 1 && ka % 4 == 0 && fast && nb > 64 && pipe_enabled() && lda < (1 << 20) && ldb < (1 << 20) && ldz < (1 << 20))) return false;
  if (workspace_floats < 0) return true;
  // the split plan of gemm_tn for this product against a workspace of that size (the window condition depends on it):
  // rows_per_split x the largest pitch must stay below 2^28 floats' worth of descriptor window
  const int64_t tiles = (int64_t)((ka + BM - 1) / BM) * ((nb + 127) / 128), slab = (int64_t)ka * nb;
  int64_t splits = 1;
  if (workspace_floats > 0 && tiles < 256) {
    splits = (256 + tiles - 1) / tiles;
    const int64_t by_rows = (m + 4 * BK - 1) / (4 * BK);
    if (splits > by_rows) splits = by_rows;
    if (splits * slab > workspace_floats) splits = workspace_floats / slab;
    if (splits < 1) splits = 1;
  }
  int64_t rps = (m + splits - 1) / splits;
  rps = (rps + BK - 1) / BK * BK;
  const int64_t ldm = lda > ldb ? (lda > ldz ? lda : ldz) : (ldb > ldz ? ldb : ldz);
  return rps * ldm < (1 << 28);
}

// glnn_gemm_tn_f32 whose final sums may be left to the fused Adam launch (glnn::PendingFolds): with `defer` the fold launch of a
// split reduction is skipped and *defer describes the slabs (same order as the fold kernel that would have run: four interleaved
// lanes from 16 splits on, k ascending below); with `defer_colsum` the second stage of col_sum_a likewise.  *used_floats = the
// workspace prefix that then has to stay untouched until Adam has run.
int glnn::gemm_tn(const float* a, int64_t lda, int64_t m, int ka, const float* b, int64_t ldb,
                  const int64_t* b_rows, const float* b_scale, const float* b_shift, float drop_p,
                  uint32_t drop_seed, int nb, float* c, int64_t ldc, float* col_sum_a, float* workspace, int64_t workspace_floats, void* stream,
                  GradFold* defer, GradFold* defer_colsum, int64_t* used_floats, int64_t plan_floats, const glnn::BnApplyA* bn) {
  if (defer) *defer = {c, nullptr, 0, 0, 0};
  if (defer_colsum) *defer_colsum = {col_sum_a, nullptr, 0, 0, 0};
  if (used_floats) *used_floats = 0;
  GLNN_REQUIRE(a && b && c, "glnn_gemm_tn_f32: null pointer");
  GLNN_REQUIRE(m >= 1 && ka >= 1 && nb >= 1, "glnn_gemm_tn_f32: bad sizes");
  GLNN_REQUIRE(lda >= ka && ldb >= nb && ldc >= nb, "glnn_gemm_tn_f32: leading dimension too small");
  GLNN_REQUIRE((b_scale == nullptr) == (b_shift == nullptr), "glnn_gemm_tn_f32: b_scale and b_shift go together");
  GLNN_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || b_scale), "glnn_gemm_tn_f32: drop_p in [0,1) and needs b_scale/b_shift");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  GemmTnArgs g;
  g.drop_thr = glnn::drop_threshold(drop_p); g.drop_seed = drop_seed; g.drop_scale = 1.0f / (1.0f - drop_p);
  g.a = a; g.lda = lda; g.m = m; g.ka = ka;
  g.b = b; g.ldb = ldb; g.b_rows = b_rows; g.b_scale = b_scale; g.b_shift = b_shift; g.nb = nb;
  g.a_vec = (lda % 4 == 0) && glnn::aligned16(a);
  g.b_vec = (ldb % 4 == 0) && glnn::aligned16(b);
  g.ax_z = nullptr; g.ax_ldz = 0; g.ax_alpha = g.ax_beta = g.ax_gamma = nullptr;
  g.ax_p1 = g.ax_p2 = nullptr; g.ax_nparts = 0; g.ax_inv_rows = 0.f; g.ax_bn_gamma = g.ax_bn_mean = g.ax_bn_rstd = nullptr;
  g.ax_dgamma = g.ax_dbeta = g.ax_colsum = nullptr;
  if (bn) {
    // A = alpha * a + beta * z + gamma on the staged pieces of the pipelined kernel: that kernel's shapes only (checked below)
    const bool parts = bn->p1 != nullptr;
    GLNN_REQUIRE(bn->z && (parts ? (bn->p2 && bn->nparts >= 1 && bn->bn_gamma && bn->bn_mean && bn->bn_rstd && bn->dgamma && bn->dbeta)
                                 : (bn->alpha && bn->beta && bn->gamma)), "glnn::gemm_tn: incomplete BnApplyA");
    if (!glnn::gemm_tn_takes_bn(a, lda, m, ka, b, ldb, nb, bn->z, bn->ldz) || b_rows || b_scale || col_sum_a)
      return GLNN_ERR_UNSUPPORTED;
    if (parts ? !(glnn::aligned16(bn->p1) && glnn::aligned16(bn->p2) && glnn::aligned16(bn->bn_gamma) && glnn::aligned16(bn->bn_mean) &&
                  glnn::aligned16(bn->bn_rstd) && glnn::aligned16(bn->dgamma) && glnn::aligned16(bn->dbeta) &&
                  (!bn->colsum || glnn::aligned16(bn->colsum)))
              : !(glnn::aligned16(bn->alpha) && glnn::aligned16(bn->beta) && glnn::aligned16(bn->gamma)))
      return GLNN_ERR_UNSUPPORTED;
    g.ax_z = bn->z; g.ax_ldz = bn->ldz; g.ax_alpha = bn->alpha; g.ax_beta = bn->beta; g.ax_gamma = bn->gamma;
    if (parts) {
      g.ax_alpha = g.ax_beta = g.ax_gamma = nullptr;
      g.ax_p1 = bn->p1; g.ax_p2 = bn->p2; g.ax_nparts = bn->nparts; g.ax_inv_rows = 1.0f / (float)bn->rows; g.ax_bn_gamma = bn->bn_gamma;
      g.ax_bn_mean = bn->bn_mean; g.ax_bn_rstd = bn->bn_rstd; g.ax_dgamma = bn->dgamma; g.ax_dbeta = bn->dbeta; g.ax_colsum = bn->colsum;
    }
  }
  int bnt = nb > 64 ? 128 : 64;
  int gi = (ka + BM - 1) / BM, gj = (nb + bnt - 1) / bnt;
  const bool fast = g.a_vec && g.b_vec && lda >= ((ka + 3) & ~3) && ldb >= ((nb + 3) & ~3) &&
                    (!b_scale || (nb % 4 == 0 && glnn::aligned16(b_scale) && glnn::aligned16(b_shift)));
  // the hand-scheduled 128 x 128 kernel: plain operands whose per-split byte offsets fit a 2 GiB descriptor window
  const bool pipe_shape = fast && bnt == 128 && !b_scale && !b_rows && pipe_enabled() && (lda > ldb ? lda : ldb) < (1 << 20);
  // latency regime (see gemm_kernel_fast): a few dozen output tiles -> 64 x 64 tiles, a quarter of the per-wave MFMA chain --
  // unless the reduction is long enough for the pipelined kernel's deeper k-loop to pay (>= 2048 rows)
  // -- nor when a WIDE a is streamed over many rows (penn94's GCN: 4814 x 64 over 41554 rows, 0.8 GB): 64-column tiles read 256-byte
  // pieces of 19 KB rows (1.45 TB/s), 128-column tiles 2.2 TB/s (552 -> 369 us)
  const bool small = !bn && tn_small_regime(fast, gi * gj, pipe_shape, m, ka);
  if (small) { bnt = 64; gi = (ka + 63) / 64; gj = (nb + 63) / 64; }
  // split the reduction over m so that the launch has >= ~256 workgroups (one per CU) when the output is small
  int splits = 1;
  const int64_t slab = (int64_t)ka * nb;
  const int64_t colsum_need = col_sum_a ? (int64_t)64 * ka : 0;
  // 64 x 64 tiles: four workgroups fit a CU, and a k-tile step is load-latency bound.  The pipelined kernel keeps its MFMA rate
  // with ONE workgroup per CU, so it splits only up to 256 workgroups: half the slab traffic of 512 (MLP3w8's 2048 x 2048
  // gradient needs no split and no fold at all: 245 -> 234 us; 60000 x 256 x 128: 97 -> 66 us; 1024 workgroups: 162 us)
  const int wg_target = small ? 1024 : (pipe_shape ? 256 : 512);
  const int min_ktiles = small ? 2 : 4;
  if (workspace && gi * gj < wg_target) {
    splits = (wg_target + gi * gj - 1) / (gi * gj);
    const int64_t max_by_rows = (m + min_ktiles * BK - 1) / (min_ktiles * BK);      // at least min_ktiles k-tiles per split
    if (splits > max_by_rows) splits = (int)max_by_rows;
    // plan_floats: plan as if the workspace had that size (a caller that hands every problem of a step its own PART of one workspace
    // wants the split plan -- hence the bits -- of the call that gets all of it); what does not fit the real size is cut down
    const int64_t avail_plan = (plan_floats > 0 ? plan_floats : workspace_floats) - colsum_need, avail = workspace_floats - colsum_need;
    if ((int64_t)splits * slab > avail_plan) splits = (int)(avail_plan / slab);
    if ((int64_t)splits * slab > avail) splits = (int)(avail / slab);
    if (splits < 1) splits = 1;
  }
  g.splits = splits;
  int64_t rps = (m + splits - 1) / splits;
  rps = (rps + BK - 1) / BK * BK;
  g.rows_per_split = rps;
  splits = (int)((m + rps - 1) / rps);
  g.splits = splits;
  const bool pipe_ok = pipe_shape && !small && rps * (lda > ldb ? lda : ldb) < (1 << 28) && (!bn || rps * bn->ldz < (1 << 28));
  if (bn && !pipe_ok) return GLNN_ERR_UNSUPPORTED;        // (nothing launched; gemm_tn_takes_bn covers every condition but the workspace-dependent split)
  float* ws_partial = workspace ? workspace + colsum_need : nullptr;
  if (splits > 1) { g.c = ws_partial; g.ldc = nb; } else { g.c = c; g.ldc = ldc; }
  const int xf = !b_scale ? 0 : (g.drop_thr ? 2 : 1);
  {
    constexpr size_t smem128 = sizeof(float) * 2 * (BK * (BM + 4) + BK * (128 + 4));
    constexpr size_t smem64 = sizeof(float) * 2 * (BK * (BM + 4) + BK * (64 + 4));
    constexpr size_t smem128t = sizeof(float) * 2 * (BM * LDS_K + 128 * LDS_K);
    constexpr size_t smem64t = sizeof(float) * 2 * (BM * LDS_K + 64 * LDS_K);
    constexpr size_t smem_s = sizeof(float) * 2 * (64 * LDS_K + 64 * LDS_K);
    static int cfg[20] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
    const bool rows = b_rows != nullptr;
    const dim3 grid(gi, gj, splits);
#define GLNN_TN_LAUNCH(KERNEL_, SLOT_, SMEM_)                                   \
  do {                                                                          \
    if (cfg[SLOT_] > 0) cfg[SLOT_] = set_smem(KERNEL_, SMEM_);                  \
    if (cfg[SLOT_] != GLNN_OK) return cfg[SLOT_];                               \
    hipLaunchKernelGGL(KERNEL_, grid, dim3(256), SMEM_, st, g);                 \
  } while (0)
    if (small) {
      if (xf == 0 && !rows) GLNN_TN_LAUNCH((gemm_tn_kernel_t<64, 64, 0, false>), 14, smem_s);
      else if (xf == 1 && !rows) GLNN_TN_LAUNCH((gemm_tn_kernel_t<64, 64, 1, false>), 15, smem_s);
      else if (!rows) GLNN_TN_LAUNCH((gemm_tn_kernel_t<64, 64, 2, false>), 16, smem_s);
      else if (xf == 0) GLNN_TN_LAUNCH((gemm_tn_kernel_t<64, 64, 0, true>), 17, smem_s);
      else if (xf == 1) GLNN_TN_LAUNCH((gemm_tn_kernel_t<64, 64, 1, true>), 18, smem_s);
      else GLNN_TN_LAUNCH((gemm_tn_kernel_t<64, 64, 2, true>), 19, smem_s);
    } else if (pipe_ok) {
      // plain operands, byte offsets inside the 2 GiB descriptor windows; the last split may end inside a k-tile
      constexpr size_t smem_pipe = pipe_lds_bytes(KROW, KROW);
      static int cfg_pipe = 1;
      if (cfg_pipe > 0) cfg_pipe = set_smem(gemm_tn_kernel_pipe, smem_pipe);
      if (cfg_pipe != GLNN_OK) return cfg_pipe;
      static int cfg_pipe_ax = 1;
      if (bn && cfg_pipe_ax > 0) cfg_pipe_ax = set_smem(gemm_tn_kernel_pipe_ax, smem_pipe);
      if (bn && cfg_pipe_ax != GLNN_OK) return cfg_pipe_ax;
      if (bn) hipLaunchKernelGGL(gemm_tn_kernel_pipe_ax, grid, dim3(256), smem_pipe, st, g);
      else hipLaunchKernelGGL(gemm_tn_kernel_pipe, grid, dim3(256), smem_pipe, st, g);
    } else if (bnt == 128) {
      if (!fast) GLNN_TN_LAUNCH((gemm_tn_kernel_generic<128>), 0, smem128);
      else if (xf == 0 && !rows) GLNN_TN_LAUNCH((gemm_tn_kernel_t<BM, 128, 0, false>), 1, smem128t);
      else if (xf == 1 && !rows) GLNN_TN_LAUNCH((gemm_tn_kernel_t<BM, 128, 1, false>), 2, smem128t);
      else if (!rows) GLNN_TN_LAUNCH((gemm_tn_kernel_t<BM, 128, 2, false>), 3, smem128t);
      else if (xf == 0) GLNN_TN_LAUNCH((gemm_tn_kernel_t<BM, 128, 0, true>), 8, smem128t);
      else if (xf == 1) GLNN_TN_LAUNCH((gemm_tn_kernel_t<BM, 128, 1, true>), 9, smem128t);
      else GLNN_TN_LAUNCH((gemm_tn_kernel_t<BM, 128, 2, true>), 10, smem128t);
    } else {
      if (!fast) GLNN_TN_LAUNCH((gemm_tn_kernel_generic<64>), 4, smem64);
      else if (xf == 0 && !rows) GLNN_TN_LAUNCH((gemm_tn_kernel_t<BM, 64, 0, false>), 5, smem64t);
      else if (xf == 1 && !rows) GLNN_TN_LAUNCH((gemm_tn_kernel_t<BM, 64, 1, false>), 6, smem64t);
      else if (!rows) GLNN_TN_LAUNCH((gemm_tn_kernel_t<BM, 64, 2, false>), 7, smem64t);
      else if (xf == 0) GLNN_TN_LAUNCH((gemm_tn_kernel_t<BM, 64, 0, true>), 11, smem64t);
      else if (xf == 1) GLNN_TN_LAUNCH((gemm_tn_kernel_t<BM, 64, 1, true>), 12, smem64t);
      else GLNN_TN_LAUNCH((gemm_tn_kernel_t<BM, 64, 2, true>), 13, smem64t);
    }
#undef GLNN_TN_LAUNCH
  }
  int rc = glnn::check_launch("glnn_gemm_tn_f32");
  if (rc != GLNN_OK) return rc;
  if (used_floats) *used_floats = (splits > 1 ? (int64_t)splits * slab : 0) + colsum_need;
  if (splits > 1 && defer && ldc == nb) {
    const bool lanes = splits >= 16 && ((nb | ldc | slab) & 3) == 0 && glnn::aligned16(ws_partial) && glnn::aligned16(c);
    *defer = {c, ws_partial, splits, lanes ? 1 : 0, slab};
  } else if (splits > 1) {
    if (splits >= 16 && ((nb | ldc | slab) & 3) == 0 && glnn::aligned16(ws_partial) && glnn::aligned16(c)) {
      int64_t blocks = ((slab >> 2) + 63) / 64;
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(split_reduce_lanes_kernel, dim3((unsigned)blocks), dim3(256), 0, st, ws_partial, slab, splits, c, ldc, ka, nb);
    } else {
      int blocks = (int)((slab + 255) / 256);
      if (blocks > 2048) blocks = 2048;
      hipLaunchKernelGGL(split_reduce_kernel, dim3(blocks), dim3(256), 0, st, ws_partial, slab, splits, c, ldc, ka, nb);
    }
    rc = glnn::check_launch("glnn_gemm_tn_f32(reduce)");
    if (rc != GLNN_OK) return rc;
  }
  if (col_sum_a) {
    GLNN_REQUIRE(workspace && workspace_floats >= colsum_need, "glnn_gemm_tn_f32: col_sum_a needs workspace >= 64*ka floats");
    int nblk = (int)((m + 255) / 256);
    if (nblk > 64) nblk = 64;
    const int64_t rpb = (m + nblk - 1) / nblk;
    hipLaunchKernelGGL(colsum_stage1, dim3((ka + 63) / 64, nblk), dim3(256), 0, st, a, lda, m, ka, rpb, workspace);
    const int nblk_used = (int)((m + rpb - 1) / rpb);           // stage 1 writes only the blocks that own rows
    if (defer_colsum && nblk_used == nblk) *defer_colsum = {col_sum_a, workspace, nblk, 0, (int64_t)ka};
    else hipLaunchKernelGGL(colsum_stage2, dim3((ka + 255) / 256), dim3(256), 0, st, workspace, nblk, ka, col_sum_a);
    rc = glnn::check_launch("glnn_gemm_tn_f32(colsum)");
  }
  return rc;
}
