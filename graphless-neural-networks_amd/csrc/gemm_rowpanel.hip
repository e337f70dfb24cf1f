// K3r: C[m, n] = epi(A[m, k] . W[n, k]^T) for SHORT reductions (k <= 128) over MANY rows, fp32 on the gfx950 matrix cores
// (v_mfma_f32_32x32x2_f32) -- the dense projection whose input is a feature / aggregate matrix (reference models.py:45,112,138):
//   * the replicated layer-1 projection of the sharded teacher (glnn_amd/dist.py): 2.45 M x 100 x 256 on every rank, 100 M x 128 x 256
//     on the synthetic-XL shard;
//   * layer 0 of teacher training over ~0.5 M-row sampled blocks (100 -> 256);
//   * the first layer of the wide students (4096 x 100 x 2048).
// Why a kernel of its own: a 128 x 128 output tile of such a product has 3-4 k-tiles.  The tiled kernels of gemm.hip (one workgroup
// per tile: load -> 4 k-tiles -> 64 stores per lane) spend their time in the prologue and epilogue of 38 k tiles, not in the loop --
// 1.81-1.95 ms = 64-69 TF for the products shape, 69 ms = 94 TF for the XL one (profiles/bench_r04_xl_a.json), against MFMA floors of
// 0.83 / 41.7 ms and byte floors of 0.6 / 26 ms.  Here the work is turned around:
//   * a workgroup (8 waves, 2 per SIMD) owns a 128-column panel of W for its whole life: the panel is loaded ONCE into LDS
//     ([128][k + pad], zero behind k) -- no weight traffic and no weight staging in the loop at all;
//   * it then WALKS 64-row tiles of A (grid-stride): wave w multiplies the 32 x 32 block (w & 1, w >> 1) of the 64 x 128 output tile,
//     both fragment kinds are conflict-free ds_read_b128 (row stride = 4 mod 8 floats), one k-group (8) ahead of the MFMAs;
//   * software pipeline ACROSS tiles, in every wave: while tile t is multiplied, the wave requests tile t+2 from memory (k-group 0:
//     more than a tile ahead, into the second of two staging register sets), stores tile t-1 from a second accumulator (k-groups 1-4)
//     and moves tile t+1 from its staging registers into the other LDS buffer (k-group 6) -- one barrier per tile, nothing between
//     barrier and first MFMA but two LDS reads; all global traffic through per-tile buffer descriptors (no divergent branch);
//   * the two waves of a SIMD interleave their MFMA chains, so one wave's stores / LDS writes issue in the shadow of the other's.
// k order: lane half kk of an MFMA takes k = 8 kg + 4 kk + t for t = 0..3 -- the pairing and sequence of gemm.hip's kernels, so the
// results are bit-identical to theirs (zero padding adds exact zeros).
#include <type_traits>

#include "glnn_common.h"

namespace {

typedef float rp_f32x16 __attribute__((ext_vector_type(16)));

constexpr int kRpRows = 64;       // rows of A per tile
constexpr int kRpCols = 128;      // columns of W per workgroup
constexpr int kRpThreads = 512;   // 8 waves: (row block w & 1) x (column block w >> 1)
constexpr int kRpPieces = 4;      // float4 per thread and A tile: 64 rows x <= 32 float4 / 512 threads

struct RpArgs {
  const float* a; int64_t lda; int64_t m; int k;
  const float* w; int64_t ldw; int n;
  const float* ep_scale; const float* ep_shift; int relu;
  float* c; int64_t ldc;
  int64_t tiles;                  // ceil(m / 64)
  // optional column statistics of C (BatchNorm training): workgroup (x, y) leaves, for each of its 128 columns, the count / mean / M2 of
  // the rows it wrote at st_*[x * n + col] -- accumulated in registers along its walk, from the very values it stores
  float* st_cnt; float* st_mean; float* st_m2;
};

__device__ __forceinline__ float4 rp_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int KG, bool STATS>     // k-groups of 8: 8 (KG - 1) < k <= 8 KG, k % 4 == 0; KG >= 5 (the side work is spread over k-groups 0 .. 4)
__global__ __launch_bounds__(kRpThreads) void gemm_rowpanel_kernel(const RpArgs g) {
  constexpr int KP = 8 * KG;
  constexpr int KS = KP + 4;      // LDS row stride (floats): KS / 4 odd -> the 16 lanes of a ds_read_b128 group hit 16 different 16-byte slots
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Wp = lds;                               // [128][KS]
  float* As = lds + kRpCols * KS;                // [2][64][KS]
  constexpr int A_TILE = kRpRows * KS;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rb = wave & 1, cb = wave >> 1;
  const int li = lane & 31, kk = lane >> 5;
  const int n0 = blockIdx.y * kRpCols;
  const int kv = g.k >> 2;                       // float4 per row of A / W

  // ---- staging map of an A tile: piece q of this thread = float4 (row, c) of the tile, f = tid + 512 q < 64 kv.  All global traffic
  //      of the loop goes through raw buffer descriptors rebuilt per tile by the scalar unit (base = the tile's first row, size = its
  //      valid rows): rows past m and idle pieces (offset 2^31) are out of range -- loads return 0, stores are dropped -- so the loop has
  //      no divergent branch and no per-lane 64-bit address arithmetic ----
  constexpr uint32_t kOob = 0x80000000u;
  uint32_t a_voff[kRpPieces];
  int a_loff[kRpPieces];
  bool a_on[kRpPieces];
#pragma unroll
  for (int q = 0; q < kRpPieces; ++q) {
    const int f = tid + kRpThreads * q;
    const int row = f / kv, c = f - row * kv;
    a_on[q] = f < kRpRows * kv;
    a_voff[q] = a_on[q] ? (uint32_t)((row * g.lda + 4 * c) * 4) : kOob;
    a_loff[q] = a_on[q] ? row * KS + 4 * c : 2 * A_TILE + 4 * tid;      // idle pieces write a private dummy slot behind the buffers
  }
  auto rows_of = [&](int64_t tile) -> int64_t {       // valid rows of a tile (0 behind the matrix)
    const int64_t left = g.m - tile * kRpRows;
    return left < 0 ? 0 : (left > kRpRows ? kRpRows : left);
  };
  auto a_rsrc = [&](int64_t tile) {
    const int64_t v = rows_of(tile);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.a) + tile * kRpRows * g.lda, 0, v > 0 ? (int)(((v - 1) * g.lda + g.k) * 4) : 0,
                                             0x00020000);
  };
  auto c_rsrc = [&](int64_t tile) {
    const int64_t v = rows_of(tile);
    return __builtin_amdgcn_make_buffer_rsrc(g.c + tile * kRpRows * g.ldc, 0, v > 0 ? (int)(((v - 1) * g.ldc + g.n) * 4) : 0, 0x00020000);
  };
  float4 stage[2][kRpPieces];                    // two tiles in flight: set S is requested while set S ^ 1 waits for its LDS slot
  auto load_tile = [&](int set, int64_t tile) {  // global -> registers
    const __amdgpu_buffer_rsrc_t rs = a_rsrc(tile);
#pragma unroll
    for (int q = 0; q < kRpPieces; ++q) stage[set][q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, a_voff[q], 0, 0));
  };
  auto store_tile_lds = [&](int set, int buf) {  // registers -> LDS
#pragma unroll
    for (int q = 0; q < kRpPieces; ++q) *reinterpret_cast<float4*>(As + (a_on[q] ? buf * A_TILE : 0) + a_loff[q]) = stage[set][q];
  };

  // ---- prologue: zero the columns behind k (one float4 per row, KP - k is 0 or 4), W panel -> LDS, tile 0 -> buffer 0 ----
  if (g.k < KP) {
    for (int r = tid; r < kRpCols + 2 * kRpRows; r += kRpThreads)
      *reinterpret_cast<float4*>(lds + r * KS + g.k) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int f = tid; f < kRpCols * kv; f += kRpThreads) {
    const int row = f / kv, c = f - row * kv;
    int ng = n0 + row;
    if (ng > g.n - 1) ng = g.n - 1;              // columns past n: a valid row re-read, never stored
    *reinterpret_cast<float4*>(Wp + row * KS + 4 * c) = rp_ld4(g.w + (int64_t)ng * g.ldw + 4 * c);
  }
  const int64_t first = blockIdx.x, stride = gridDim.x;
  const int64_t n_my = first < g.tiles ? (g.tiles - first + stride - 1) / stride : 0;     // tiles this workgroup walks
  if (n_my == 0) return;
  load_tile(0, first);
  store_tile_lds(0, 0);
  load_tile(1, first + stride);                  // tile 1 (nothing, if the walk has one tile): in flight across the first tile
  __syncthreads();

  // per-column epilogue constants of this wave's 32 columns (fixed for the whole walk)
  const int col = n0 + cb * 32 + li;
  const bool col_ok = col < g.n;
  const float es = (col_ok && g.ep_scale) ? g.ep_scale[col] : 1.f;
  const float eh = (col_ok && g.ep_shift) ? g.ep_shift[col] : 0.f;
  const float* bp = Wp + (cb * 32 + li) * KS + kk * 4;
  const int a_frag = (rb * 32 + li) * KS + kk * 4;
  // C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5): element r of the accumulator goes to
  // tile row  rb*32 + 4*kk  +  8*(r >> 2) + (r & 3)  -- the first part per lane, the second one uniform (the store's scalar offset)
  const uint32_t c_voff = col_ok ? (uint32_t)(((rb * 32 + 4 * kk) * g.ldc + col) * 4) : kOob;
  const uint32_t ldc4 = (uint32_t)(g.ldc * 4);

  rp_f32x16 acc[2];
  // STATS: per lane, the sums of (v - shift) and (v - shift)^2 over the 16 values of its column it stores per tile; shift = the first value
  // the lane stores (within a few sigma of the column mean: the one-pass variance loses a few bits, not the digits a zero shift would).
  // `rows_ok` = valid rows of the stored tile counted from this lane's first row (rb*32 + 4*kk): only the matrix's ragged last tile masks.
  float st_shift = 0.f, st_s1 = 0.f, st_s2 = 0.f;
  auto store_quarter = [&](const rp_f32x16& av, const __amdgpu_buffer_rsrc_t rs, int q0, bool first, int rows_ok) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float v = fmaf(av[4 * q0 + t], es, eh);
      if (g.relu) v = fmaxf(v, 0.f);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, c_voff, (uint32_t)(8 * q0 + t) * ldc4, 0);
      if (STATS) {
        if (q0 == 0 && t == 0) st_shift = first ? v : st_shift;
        const float dv = (8 * q0 + t < rows_ok) ? v - st_shift : 0.f;
        st_s1 += dv;
        st_s2 = fmaf(dv, dv, st_s2);
      }
    }
  };

  // one tile out of LDS buffer CUR into accumulator set CUR; `it` = its index in this workgroup's walk
  auto tile_step = [&](auto cur_, int64_t it) {
    constexpr int CUR = decltype(cur_)::value;
    const int64_t tile = first + it * stride;
    // the previous tile leaves from the other accumulator set (walk start: a descriptor of size 0 -- every store is dropped)
    const __amdgpu_buffer_rsrc_t prev = c_rsrc(it > 0 ? tile - stride : g.tiles);
    const float* ap = As + CUR * A_TILE + a_frag;
    float4 af = rp_ld4(ap), bf = rp_ld4(bp);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[CUR][r] = 0.f;
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
      float4 an = af, bn = bf;
      if (kg + 1 < KG) {
        an = rp_ld4(ap + (kg + 1) * 8);
        bn = rp_ld4(bp + (kg + 1) * 8);
      }
      acc[CUR] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc[CUR], 0, 0, 0);
      acc[CUR] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc[CUR], 0, 0, 0);
      acc[CUR] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc[CUR], 0, 0, 0);
      acc[CUR] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc[CUR], 0, 0, 0);
      // ---- side work, in the shadow of the MFMAs (this wave's and its SIMD neighbour's) ----
      if (kg == 0) load_tile(CUR, tile + 2 * stride);            // tile it+2: requested more than a tile ahead, into staging set CUR
      if (kg >= 1 && kg < 5) store_quarter(acc[CUR ^ 1], prev, kg - 1, it == 1, it > 0 ? 64 : 0);   // (never the ragged tile; walk start: nothing to count)
      if (kg == (KG > 6 ? 6 : KG - 1)) store_tile_lds(CUR ^ 1, CUR ^ 1);   // tile it+1 (requested during tile it-1): staging set CUR^1 -> the other buffer
      // (all unconditional: behind the walk's end the loads are out of range and the LDS write goes to a buffer nobody reads)
      __builtin_amdgcn_sched_barrier(0);                         // keep the side work of a k-group with its MFMAs
      af = an; bf = bn;
    }
    __syncthreads();      // buffer CUR ^ 1 is complete, and everybody is done reading buffer CUR
  };

  for (int64_t it = 0; it < n_my; it += 2) {
    tile_step(std::integral_constant<int, 0>{}, it);
    if (it + 1 < n_my) tile_step(std::integral_constant<int, 1>{}, it + 1);
  }
  // the last tile's stores
  const int64_t last = n_my - 1;
  const __amdgpu_buffer_rsrc_t rl = c_rsrc(first + last * stride);
  const int rows_last = (int)rows_of(first + last * stride) - (rb * 32 + 4 * kk);      // may be <= 0: nothing of this lane's rows is valid
  if (last & 1) {
#pragma unroll
    for (int q0 = 0; q0 < 4; ++q0) store_quarter(acc[1], rl, q0, last == 0, rows_last);
  } else {
#pragma unroll
    for (int q0 = 0; q0 < 4; ++q0) store_quarter(acc[0], rl, q0, last == 0, rows_last);
  }
  if (STATS) {
    // lane -> (count, mean, M2) of its values; the four holders of a column (2 lane halves x 2 row-block waves) are combined in double
    // (Chan) by one thread per column.  Accumulator element r sits at row offset 8 (r >> 2) + (r & 3): count the valid ones of the last tile
    int n_last = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) n_last += (8 * (r >> 2) + (r & 3) < rows_last) ? 1 : 0;
    const float cnt = (float)(16 * (n_my - 1) + n_last);
    const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
    const float mean_l = st_shift + st_s1 * inv;
    const float m2_l = fmaxf(st_s2 - st_s1 * st_s1 * inv, 0.f);
    // (the walk's last barrier is behind every LDS read: the buffers are free)
    float* red = lds;                              // [3][8 waves][64 lanes]
    red[wave * 64 + lane] = cnt;
    red[512 + wave * 64 + lane] = mean_l;
    red[1024 + wave * 64 + lane] = m2_l;
    __syncthreads();
    if (tid < kRpCols) {
      const int cbk = tid >> 5, lik = tid & 31, colk = n0 + tid;
      double n = 0.0, mu = 0.0, m2 = 0.0;
#pragma unroll
      for (int h = 0; h < 4; ++h) {                // (row block 0, half 0), (0, 1), (1, 0), (1, 1): rows ascending within a tile
        const int src = (cbk * 2 + (h >> 1)) * 64 + lik + 32 * (h & 1);
        const double nb = red[src], mb = red[512 + src], qb = red[1024 + src];
        if (nb > 0.0) {
          const double nt = n + nb, dl = mb - mu;
          mu += dl * nb / nt;
          m2 += qb + dl * dl * n * nb / nt;
          n = nt;
        }
      }
      if (colk < g.n) {
        const int64_t o = (int64_t)blockIdx.x * g.n + colk;
        g.st_cnt[o] = (float)n;
        g.st_mean[o] = (float)mu;
        g.st_m2[o] = (float)m2;
      }
    }
  }
}

template <int KG, bool STATS>
int launch_rowpanel_t(const RpArgs& g, int grid_x, int panels, hipStream_t st) {
  constexpr size_t smem = sizeof(float) * ((size_t)(kRpCols + 2 * kRpRows) * (8 * KG + 4) + 4 * kRpThreads);      // + the idle pieces' dummy slots
  static int configured = 0;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_rowpanel_kernel<KG, STATS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return glnn::fail(GLNN_ERR_HIP, "gemm_rowpanel: hipFuncSetAttribute(max dynamic LDS=%zu) failed", smem);
    configured = 1;
  }
  hipLaunchKernelGGL((gemm_rowpanel_kernel<KG, STATS>), dim3((unsigned)grid_x, (unsigned)panels), dim3(kRpThreads), smem, st, g);
  return glnn::check_launch("glnn_gemm_f32(rowpanel)");
}
template <int KG>
int launch_rowpanel(const RpArgs& g, int grid_x, int panels, hipStream_t st) {
  return g.st_mean ? launch_rowpanel_t<KG, true>(g, grid_x, panels, st) : launch_rowpanel_t<KG, false>(g, grid_x, panels, st);
}

}  // namespace

// GLNN_ERR_UNSUPPORTED = nothing launched (the caller takes the tiled kernels): plain float4-addressable operands, W [n, k] with
// 36 <= k <= 128 (k % 4 == 0) -- from k = 129 on the pipelined k loop of gemm.hip amortises its prologue -- and enough rows that every
// workgroup walks >= 2 tiles of a full-chip grid.
int glnn::gemm_rowpanel(const float* a, int64_t lda, int64_t m, int k, const float* w, int64_t ldw, int n, const float* ep_scale,
                        const float* ep_shift, int relu, float* c, int64_t ldc, void* stream, glnn::ColStats* cs) {
  if (k < 36 || k > 128 || (k & 3) || m < 2048 || n < 96) return GLNN_ERR_UNSUPPORTED;
  if ((lda & 3) || (ldw & 3) || lda < k || ldw < k || ldc < n || !glnn::aligned16(a) || !glnn::aligned16(w)) return GLNN_ERR_UNSUPPORTED;
  if (lda >= (1 << 22) || ldc >= (1 << 22)) return GLNN_ERR_UNSUPPORTED;      // a 64-row tile must fit a 2 GiB buffer window
  RpArgs g;
  g.a = a; g.lda = lda; g.m = m; g.k = k; g.w = w; g.ldw = ldw; g.n = n; g.ep_scale = ep_scale; g.ep_shift = ep_shift; g.relu = relu;
  g.c = c; g.ldc = ldc; g.tiles = (m + kRpRows - 1) / kRpRows;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int kg = (k + 7) / 8;
  const int panels = (n + kRpCols - 1) / kRpCols;
  if (panels > 65535) return GLNN_ERR_UNSUPPORTED;
  // one workgroup per CU (the W panel + two A buffers take 108-132 KB of the CU's 160 KB LDS): 256 workgroups in all, the `panels`
  // workgroups of a row strip x on the same XCD (linear id = y * grid_x + x, dispatched round-robin over the 8 XCDs: grid_x % 8 == 0),
  // where they share the strip's rows of A in L2
  int64_t gx = 256 / panels;
  if (gx < 8) gx = 8;
  gx &= ~(int64_t)7;
  while (gx > 8 && g.tiles < 2 * gx) gx -= 8;
  if (g.tiles < 2 * gx) return GLNN_ERR_UNSUPPORTED;                            // too few tiles per workgroup to pay for the panel load
  g.st_cnt = g.st_mean = g.st_m2 = nullptr;
  if (cs && !relu && cs->ws && cs->ws_floats >= 3 * gx * (int64_t)n) {          // (statistics of an activated output are nobody's BatchNorm input)
    g.st_cnt = cs->ws;
    g.st_mean = cs->ws + gx * (int64_t)n;
    g.st_m2 = cs->ws + 2 * gx * (int64_t)n;
  }
  const auto launch = [&]() -> int {
  switch (kg) {
    case 5: return launch_rowpanel<5>(g, (int)gx, panels, st);
    case 6: return launch_rowpanel<6>(g, (int)gx, panels, st);
    case 7: return launch_rowpanel<7>(g, (int)gx, panels, st);
    case 8: return launch_rowpanel<8>(g, (int)gx, panels, st);
    case 9: return launch_rowpanel<9>(g, (int)gx, panels, st);
    case 10: return launch_rowpanel<10>(g, (int)gx, panels, st);
    case 11: return launch_rowpanel<11>(g, (int)gx, panels, st);
    case 12: return launch_rowpanel<12>(g, (int)gx, panels, st);
    case 13: return launch_rowpanel<13>(g, (int)gx, panels, st);
    case 14: return launch_rowpanel<14>(g, (int)gx, panels, st);
    case 15: return launch_rowpanel<15>(g, (int)gx, panels, st);
    case 16: return launch_rowpanel<16>(g, (int)gx, panels, st);
    default: return GLNN_ERR_UNSUPPORTED;
  }
  };
  const int rc = launch();
  if (rc == GLNN_OK && g.st_mean) {
    cs->ws_cnt = g.st_cnt; cs->ws_mean = g.st_mean; cs->ws_m2 = g.st_m2;
    cs->nparts = (int)gx; cs->chunk_rows = 0; cs->done = 1;
  }
  return rc;
}
