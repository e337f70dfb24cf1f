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


// ---------------------------------------------------------------------------------------------
// K3w (round 5): the WAVE-WALK form.  Same product, same k order per output element (bit-identical results), but no workgroup
// tile, no LDS staging of A and no barrier in the walk at all:
//   * a workgroup still shares ONE thing: the 128-column panel of W in LDS, loaded once (one barrier, before the walk);
//   * every WAVE walks its own sequence of 32-row tiles of A and owns the whole 32 x 128 output tile (four 32 x 32 MFMA blocks:
//     every A value feeds four MFMAs, every ds_read_b128 of a W fragment feeds four);
//   * A never touches LDS: lane (li, kk) of the 32x32x2 MFMA needs A[row li][8 kg + 4 kk .. +3] -- one float4 per k-group, loaded
//     straight into the MFMA's operand registers through a per-tile buffer descriptor (rows past m and the k tail read 0).  The
//     registers ROLL: the load of k-group kg of tile t+1 is issued right behind the last MFMA that reads k-group kg of tile t, so a
//     load has a whole tile (13 k 312 MFMA cycles) to land and one register set serves both tiles.  The W fragments roll the same
//     way one k-group ahead (16 registers);
//   * the accumulators (AGPRs) leave at the end of the tile -- 64 x {read, epilogue, dword store} -- under the MFMAs of the SIMD's other
//     wave (one wave alone saturates the matrix cores: its dependent MFMAs are four slots apart); the two waves of a SIMD start half a
//     tile apart so that their store phases never coincide;
//   * the instruction stream of the walk is stated slot by slot (asm volatile, hand-counted s_waitcnt, the pipe_mainloop treatment of
//     gemm.hip): per k-group 16 MFMAs (t-major over the four column blocks: dependent MFMAs are four slots apart), 4 ds_read_b128 and
//     1 buffer_load_dwordx4; nothing waits on anything younger than a k-group (LDS) or most of a tile (global).
// What it buys (profiles/r05_k3_power.txt, r05_rowwalk_times.txt): 15 % fewer cycles than the round-4 form -- and 5 % less time, because the
// product runs AT the part's 1400 W power cap: without memory traffic this loop takes 972 us at 2384 MHz and 733 W (134 TF on the padded
// K = 104); its stores add 350 W, its loads 617 W, and with both the socket sits at 1363-1378 W with the engine clock pulled to ~2.1 GHz:
// 1150-1170 us = 107-109 TF sustained on 2,449,029 x 100 x 256 (round-4 form: same cap at 2.32 GHz, 1229-1239 us), 117-120 TF at K = 128.
// ---------------------------------------------------------------------------------------------
typedef float rw_f32x4 __attribute__((ext_vector_type(4)));
typedef int rw_i32x4 __attribute__((ext_vector_type(4)));

template <class F, int... I>
__device__ __forceinline__ void rw_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void rw_static_for(F&& f) {
  rw_static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ rw_i32x4 rw_rsrc(const float* base, int64_t bytes) {      // raw buffer descriptor over [base, base + bytes)
  const uint64_t b = reinterpret_cast<uint64_t>(base);
  rw_i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)b);
  r.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)((b >> 32) & 0xFFFFu));
  r.z = __builtin_amdgcn_readfirstlane((int)(bytes < 0 ? 0 : bytes));
  r.w = 0x00020000;
  return r;
}

#define RW_MFMA(ACC_, A_, B_) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(ACC_) : "v"(A_), "v"(B_))
#define RW_MFMA0(ACC_, A_, B_) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=a"(ACC_) : "v"(A_), "v"(B_))
#define RW_DS_READ(DST_, ADDR_, OFF_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST_) : "v"(ADDR_), "n"(OFF_) : "memory")
#define RW_BLOAD(DST_, VOFF_, RSRC_, OFF_) \
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "=v"(DST_) : "v"(VOFF_), "s"(RSRC_), "n"(OFF_) : "memory")
#define RW_BSTORE(VAL_, VOFF_, RSRC_, SOFF_, OFF_) \
  asm volatile("buffer_store_dword %0, %1, %2, %3 offen offset:%4" : : "v"(VAL_), "v"(VOFF_), "s"(RSRC_), "s"(SOFF_), "n"(OFF_) : "memory")

constexpr int kRwRows = 32;       // rows of A per wave tile
#ifndef GLNN_RW_LOAD_GROUP
#define GLNN_RW_LOAD_GROUP 4
#endif
constexpr int kRwLoadGroup = GLNN_RW_LOAD_GROUP;      // (1 / 2 / 4 / 16 measured 1170 / 1165 / 1152 / 1221 us on 2.45 M x 100 x 256: profiles/r05_k3_power.txt)
constexpr int kRwWaves = 8;       // waves per workgroup: two per SIMD (4: no neighbour to hide the store phase; 12: measured equal, and KG = 16 spilled)

template <int KG, bool RELU>
__global__ __launch_bounds__(64 * kRwWaves) void gemm_rowwalk_kernel(const RpArgs g) {
  constexpr int KP = 8 * KG;
  constexpr int KS = KP + 4;      // LDS row stride of the W panel (floats): KS / 4 odd -> conflict-free ds_read_b128
  extern __shared__ __attribute__((aligned(16))) float lds[];      // W panel [128][KS]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kk = lane >> 5;
  const int n0 = blockIdx.y * kRpCols;
  const int kv = g.k >> 2;
  constexpr uint32_t kOob = 0x80000000u;

  const int64_t wtiles = (g.m + kRwRows - 1) / kRwRows;
  const int64_t first = (int64_t)blockIdx.x * kRwWaves + wave, stride = (int64_t)gridDim.x * kRwWaves;
  // (tile counts fit 32 bits: lda / ldc < 2^22 bounds m far below 2^36; the scalar unit has no 64-bit divide)
  const int n_my = first < wtiles ? __builtin_amdgcn_readfirstlane((int)((uint32_t)(wtiles - first + stride - 1) / (uint32_t)stride)) : 0;
  auto rows_of = [&](int64_t tile) -> int {          // valid rows of a wave tile (0 behind the matrix)
    const int64_t left = g.m - tile * kRwRows;
    return (int)(left < 0 ? 0 : (left > kRwRows ? kRwRows : left));
  };
  auto a_rsrc = [&](int64_t tile) {
#ifdef GLNN_RW_NO_LOADS      // probe builds only (scripts/rowwalk_ab.sh): every fragment load is out of range -- no memory access, zeros
    const int v = 0 * rows_of(tile);
#else
    const int v = rows_of(tile);
#endif
    return rw_rsrc(g.a + (v > 0 ? tile : 0) * kRwRows * g.lda, v > 0 ? ((int64_t)(v - 1) * g.lda + g.k) * 4 : 0);
  };
  auto c_rsrc = [&](int64_t tile) {
#ifdef GLNN_RW_NO_STORES     // probe builds only: every store is out of range and dropped
    const int v = 0 * rows_of(tile);
#else
    const int v = rows_of(tile);
#endif
    return rw_rsrc(g.c + (v > 0 ? tile : 0) * kRwRows * g.ldc, v > 0 ? ((int64_t)(v - 1) * g.ldc + g.n) * 4 : 0);
  };

  // lane (li, kk) reads A[row li][8 kg + 4 kk ..]: byte offset below + 32 kg as the instruction's immediate.  The last k-group's upper
  // half lies behind k when k = 8 KG - 4: those lanes are out of range (offset 2^31) and read 0 -- not the next row's first floats.
  const uint32_t a_voff = (uint32_t)((li * g.lda + 4 * kk) * 4);
  const uint32_t a_voff_last = (8 * (KG - 1) + 4 * kk < g.k) ? a_voff : kOob;
  rw_f32x4 af[KG];                 // A fragments of the current tile, k-group by k-group (rolling: see above)
  rw_f32x4 fb[4];                  // W fragments of the current k-group, one per column block (rolling)

  // ---- prologue: the first tile's A fragments are requested, then the W panel goes to LDS (zero behind k), one barrier ----
  {
    const rw_i32x4 rs0 = a_rsrc(n_my > 0 ? first : wtiles);
    rw_static_for<KG>([&](auto kg_) {
      constexpr int kg = decltype(kg_)::value;
      (void)af; (void)a_voff; (void)a_voff_last; (void)rs0;
      if constexpr (kg == KG - 1) RW_BLOAD(af[kg], a_voff_last, rs0, kg * 32);
      else RW_BLOAD(af[kg], a_voff, rs0, kg * 32);
    });
  }
  if (g.k < KP) {
    for (int r = tid; r < kRpCols; r += 64 * kRwWaves) *reinterpret_cast<float4*>(lds + r * KS + g.k) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int f = tid; f < kRpCols * kv; f += 64 * kRwWaves) {
    const int row = f / kv, c = f - row * kv;
    int ng = n0 + row;
    if (ng > g.n - 1) ng = g.n - 1;              // columns past n: a valid row re-read, never stored
    *reinterpret_cast<float4*>(lds + row * KS + 4 * c) = rp_ld4(g.w + (int64_t)ng * g.ldw + 4 * c);
  }
  __syncthreads();

  // per-column constants of the four column blocks (fixed for the whole walk)
  float es[4], eh[4];
  uint32_t c_voff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = n0 + 32 * j + li;
    const bool ok = col < g.n;
    es[j] = (ok && g.ep_scale) ? g.ep_scale[col] : 1.f;
    eh[j] = (ok && g.ep_shift) ? g.ep_shift[col] : 0.f;
    // C/D map of the 32x32 MFMA: col = lane & 31, row = 4 (lane >> 5) + 8 (r >> 2) + (r & 3): the lane part here, the r part uniform
    c_voff[j] = ok ? (uint32_t)((4 * kk * g.ldc + col) * 4) : kOob;
  }
  const uint32_t ldc4 = (uint32_t)(g.ldc * 4);
  const uint32_t bp = (uint32_t)(uintptr_t)lds + (uint32_t)((li * KS + 4 * kk) * 4);      // W fragment of column block j, k-group kg: + (32 j KS + 8 kg) * 4

  rp_f32x16 acc[4];
  // (the compiler's own loads above are consumed here, so that its s_waitcnt for them is not placed inside the walk)
#pragma unroll
  for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(es[j]), "+v"(eh[j]), "+v"(c_voff[j]));
  // the W fragments of k-group 0 and everything the prologue requested: in registers before the walk starts
  rw_static_for<4>([&](auto j_) {
    constexpr int j = decltype(j_)::value;
    (void)fb; (void)bp;
    RW_DS_READ(fb[j], bp, (32 * j * KS) * 4);
  });
  rw_static_for<KG>([&](auto kg_) {
    constexpr int kg = decltype(kg_)::value;
    (void)af;
    if constexpr (kg == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(af[kg]) : : "memory");
    else asm volatile("" : "+v"(af[kg]) : : "memory");
  });
  asm volatile("" : "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]) : : "memory");
#ifndef GLNN_RW_NO_STAGGER
  // the two waves of a SIMD (w, w + 4) start half a tile apart: a wave's store phase then lies under its neighbour's MFMAs for the
  // whole walk (in lockstep both would store at the same time and the matrix cores would idle).  One wave alone keeps them busy.
  if (wave >= 4 && n_my >= 4) {
    constexpr int part = KG * 16 / 2;                     // half a tile's MFMA time, in units of 64 clocks
    __builtin_amdgcn_s_sleep(part > 127 ? 127 : part);
  }
#endif

  // one tile: KG x 16 MFMA slots with the rolling fragment loads, then the tile leaves (epilogue, 64 stores)
  auto step = [&](int it) {
    const int64_t tile = first + (int64_t)it * stride;
    const rw_i32x4 rs_next = a_rsrc(it + 1 < n_my ? tile + stride : wtiles);
    const rw_i32x4 rs_c = c_rsrc(tile);
    rw_static_for<KG * 16>([&](auto s_) {
      constexpr int kg = decltype(s_)::value / 16, mm = decltype(s_)::value % 16;
      constexpr int t = mm / 4, j = mm % 4;
      (void)af; (void)fb; (void)acc; (void)bp; (void)a_voff; (void)a_voff_last; (void)rs_next;
      if constexpr (mm == 0) {
        // this k-group's A fragment was requested a tile ago: KG - 1 loads and the 64 stores of a tile were issued behind it (memory
        // operations complete in order; the counter saturates at 63).  (First tile: the prologue waited for everything.)
        static_assert(KG - 1 + 64 >= 63, "vmcnt");
        asm volatile("s_waitcnt vmcnt(63)" : "+v"(af[kg]) : : "memory");
      }
      if constexpr (t == 0) {
        // the four fragment reads of this k-group were issued behind slots 12..15 of the previous one, in block order
        if constexpr (j == 0) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fb[0]) : : "memory");
        if constexpr (j == 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fb[1]) : : "memory");
        if constexpr (j == 2) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(fb[2]) : : "memory");
        if constexpr (j == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[3]) : : "memory");
      }
      if constexpr (kg == 0 && t == 0) RW_MFMA0(acc[j], af[kg][t], fb[j][t]);
      else RW_MFMA(acc[j], af[kg][t], fb[j][t]);
      if constexpr (t == 3) {
        // block j's fragment has fed its last MFMA of this k-group: request the next k-group's (the next tile's first, at the end)
        constexpr int kn = (kg + 1) % KG;
        RW_DS_READ(fb[j], bp, (32 * j * KS + 8 * kn) * 4);
      }
      // the next tile's fragments of the k-groups that are done, kRwLoadGroup of them together (a 128-byte line of A holds four k-groups
      // of a row: requested back to back they meet in the vector cache instead of four separate trips to L2)
      constexpr int LG = kRwLoadGroup;
      if constexpr (mm == 15 && (kg % LG == LG - 1 || kg == KG - 1)) {
        constexpr int g0 = kg - kg % LG;
        rw_static_for<kg - g0 + 1>([&](auto q_) {
          constexpr int kq = g0 + decltype(q_)::value;
          (void)af; (void)a_voff; (void)a_voff_last; (void)rs_next;
          if constexpr (kq == KG - 1) RW_BLOAD(af[kq], a_voff_last, rs_next, kq * 32);
          else RW_BLOAD(af[kq], a_voff, rs_next, kq * 32);
        });
      }
    });
    // the tile leaves, block by block.  Block j's last MFMA is slot 204 + j; block 0's has retired when slot 207 has issued, the later
    // ones retire under the >= 48 instructions of the blocks in front of them (XDL write -> read: 18 wait states)
    asm volatile("s_nop 7" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]));
    rw_static_for<64>([&](auto e_) {
      constexpr int e = decltype(e_)::value;
      constexpr int jb = e / 16, r = e % 16;
      constexpr int roff = 8 * (r >> 2) + (r & 3);
      (void)acc; (void)es; (void)eh; (void)c_voff; (void)ldc4; (void)rs_c;
      float av;                                        // (stated as asm: keeps the read next to its store)
      asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(av) : "a"(acc[jb][r]));
      float v = fmaf(av, es[jb], eh[jb]);
      if constexpr (RELU) v = fmaxf(v, 0.f);
      const uint32_t so = (uint32_t)roff * ldc4;
      RW_BSTORE(v, c_voff[jb], rs_c, so, 0);
    });
  };

  // ONE copy of the step and no code behind the loop that touches the asm-managed registers: a control-flow join is where the compiler
  // copies registers whose asm loads / MFMAs may still be in flight.  Fragment loads past the walk's end are out of range (no access).
  for (int it = 0; it < n_my; ++it) step(it);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

// dynamic LDS above the 64 KB default needs the attribute once per kernel AND device
template <class K>
int rp_configure_lds(K kernel, size_t smem, int* configured_mask) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return glnn::fail(GLNN_ERR_NO_DEVICE, "gemm_rowpanel: no HIP device");
  if (dev < 0 || dev >= 32) dev = 31;
  if (dev != 31 && ((*configured_mask >> dev) & 1)) return GLNN_OK;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
    (void)hipGetLastError();                       // a part with less LDS: nothing launched, the caller takes the tiled kernels
    return GLNN_ERR_UNSUPPORTED;
  }
  if (dev != 31) *configured_mask |= 1 << dev;
  return GLNN_OK;
}

template <int KG, bool RELU>
int launch_rowwalk_t(const RpArgs& g, int grid_x, int panels, hipStream_t st) {
  constexpr size_t smem = sizeof(float) * (size_t)kRpCols * (8 * KG + 4);
  static int configured = 0;
  const int rc = rp_configure_lds(gemm_rowwalk_kernel<KG, RELU>, smem, &configured);
  if (rc != GLNN_OK) return rc;
  hipLaunchKernelGGL((gemm_rowwalk_kernel<KG, RELU>), dim3((unsigned)grid_x, (unsigned)panels), dim3(64 * kRwWaves), smem, st, g);
  return glnn::check_launch("glnn_gemm_f32(rowwalk)");
}
template <int KG>
int launch_rowwalk(const RpArgs& g, int grid_x, int panels, hipStream_t st) {
  return g.relu ? launch_rowwalk_t<KG, true>(g, grid_x, panels, st) : launch_rowwalk_t<KG, false>(g, grid_x, panels, st);
}

template <int KG, bool STATS>
int launch_rowpanel_t(const RpArgs& g, int grid_x, int panels, hipStream_t st) {
  constexpr size_t smem = sizeof(float) * ((size_t)(kRpCols + 2 * kRpRows) * (8 * KG + 4) + 4 * kRpThreads);      // + the idle pieces' dummy slots
  static int configured = 0;
  const int rcc = rp_configure_lds(gemm_rowpanel_kernel<KG, STATS>, smem, &configured);
  if (rcc != GLNN_OK) return rcc;
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
  // which form: the wave-walk kernel (round 5) takes the plain products; a product whose epilogue also leaves BatchNorm statistics stays on the
  // workgroup-tile kernel of round 4 (the statistics' VALU work next to the walk's asm-managed registers made the compiler spill them;
  // measured slower as well: 336 vs 309 us on 0.5 M x 100 x 256).  GLNN_GEMM_ROWPANEL=2: everything on the round-4 form (A/B, tests).
  const bool want_stats = cs && !relu && cs->ws;      // (statistics of an activated output are nobody's BatchNorm input)
  const bool walk = glnn::opts().gemm_rowpanel != 2 && !want_stats;
  // one workgroup per CU: 256 workgroups in all, the `panels` workgroups of a row strip x on the same XCD (linear id = y * grid_x + x,
  // dispatched round-robin over the 8 XCDs: grid_x % 8 == 0), where they share the strip's rows of A in L2
  int64_t gx = 256 / panels;
  if (gx < 8) gx = 8;
  gx &= ~(int64_t)7;
  const int64_t units = walk ? (m + kRwRows - 1) / kRwRows : g.tiles;        // wave tiles / workgroup tiles
  const int64_t per_wg = walk ? kRwWaves : 2;                                 // >= 1 tile per wave / >= 2 tiles per workgroup
  while (gx > 8 && units < per_wg * gx) gx -= 8;
  if (units < per_wg * gx) return GLNN_ERR_UNSUPPORTED;                       // too few tiles to pay for the panel load
  g.st_cnt = g.st_mean = g.st_m2 = nullptr;
  if (want_stats && cs->ws_floats >= 3 * gx * (int64_t)n) {
    g.st_cnt = cs->ws;
    g.st_mean = cs->ws + gx * (int64_t)n;
    g.st_m2 = cs->ws + 2 * gx * (int64_t)n;
  }
#ifdef GLNN_RP_DEV      // development builds (scripts/build_variant.sh): only the two reductions of the benchmarks, a tenth of the compile time
#define RP_CASES(FN) \
    case 13: return FN<13>(g, (int)gx, panels, st); \
    case 16: return FN<16>(g, (int)gx, panels, st);
#else
#define RP_CASES(FN) \
    case 5: return FN<5>(g, (int)gx, panels, st);   case 6: return FN<6>(g, (int)gx, panels, st);   case 7: return FN<7>(g, (int)gx, panels, st); \
    case 8: return FN<8>(g, (int)gx, panels, st);   case 9: return FN<9>(g, (int)gx, panels, st);   case 10: return FN<10>(g, (int)gx, panels, st); \
    case 11: return FN<11>(g, (int)gx, panels, st); case 12: return FN<12>(g, (int)gx, panels, st); case 13: return FN<13>(g, (int)gx, panels, st); \
    case 14: return FN<14>(g, (int)gx, panels, st); case 15: return FN<15>(g, (int)gx, panels, st); case 16: return FN<16>(g, (int)gx, panels, st);
#endif
  const auto launch = [&]() -> int {
    if (walk) {
      switch (kg) {
        RP_CASES(launch_rowwalk)
        default: return GLNN_ERR_UNSUPPORTED;
      }
    }
    switch (kg) {
      RP_CASES(launch_rowpanel)
      default: return GLNN_ERR_UNSUPPORTED;
    }
  };
#undef RP_CASES
  const int rc = launch();
  if (rc == GLNN_OK && g.st_mean) {
    cs->ws_cnt = g.st_cnt; cs->ws_mean = g.st_mean; cs->ws_m2 = g.st_m2;
    cs->nparts = (int)gx; cs->chunk_rows = 0; cs->done = 1;
  }
  return rc;
}
