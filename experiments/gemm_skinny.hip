// EXPERIMENT (round 3), not part of libglnn_hip.so: a W-resident "skinny k" GEMM for 2.4 M x 100 x 256 (the replicated layer-1
// projection of the sharded teacher).  Four forms were built, all bit-identical to the tiled kernels, none faster than them:
//   tiled gemm_kernel_fast (shipping)                                   1.81 ms  (69 TF/s; rocBLAS 1.51-1.56 ms)
//   v1 W panel in LDS, A fragments straight from global (16 B / lane)   1.70 ms  (51 KB of fragment-shaped A per CU thrash the 32 KB L1)
//   v2 W panel as B fragments in registers (208 / lane)                 (compiler parks them in AGPRs, v_accvgpr_read per operand, spills)
//   v3 W in LDS + A through a private LDS tile, 4 waves / CU            3.14 ms  (one wave per SIMD: nothing hides its own latencies)
//   v4 (this file) the same with 8 waves / CU, unpadded stride 100      2.44 ms
// and the "k_pad_plan" (run the pipelined kernel over k = 128 with a zero-padded weight copy): 1.81 vs 1.84 ms, no gain -- with 4
// k-tiles the tiled kernels are prologue / epilogue bound whatever the loop is.  One compiler trap worth keeping: an array written
// under a loop-carried condition (`if (next tile exists) stg[i] = load`) is kept in SCRATCH memory; making the load unconditional
// (re-load the last tile) brought it back into registers.
// K3s: C[m,n] = epi(A[m,k] . W[n,k]^T) for SHORT reductions (36 <= k <= 128, k % 4 == 0) over MANY rows -- the shape of a
// dense projection whose input is a feature / aggregate matrix: ogbn-products' 100-wide layer-1 aggregate -> 256 hidden
// (2.4 M x 100 x 256: the projection every rank of the sharded teacher replicates, glnn_amd/dist.py; the first layer of teacher
// training over 0.5 M-row blocks; the first layer of the students, models.py:45,112,138).
//
// Why a kernel of its own: with k = 100 a 128 x 128 output tile has 3-4 k-tiles -- the tiled kernels (gemm.hip) spend their time
// in the prologue / epilogue of 38 k tiles, not in the loop: 1.81 ms = 69 TF/s, 1.9 TB/s (rocBLAS 1.53 ms) for a product whose
// MFMA floor is 0.83 ms and whose byte floor (0.98 GB in, 2.5 GB out) is 0.65 ms.  Here
//   * the W panel of a workgroup (128 output columns x k) is loaded ONCE into LDS and stays there while the workgroup walks row
//     tiles (persistent over rows: no per-tile weight traffic, no barrier inside the loop);
//   * A goes global -> registers -> LDS in whole lines: every wave copies its own 32 rows (one contiguous 12.8 KB block when
//     lda == k) with lane-linear 16-byte loads into a PRIVATE LDS tile -- the next 32 rows are requested before the current ones
//     are multiplied and written over the tile behind the MFMAs (LDS operations of one wave are ordered: no barrier);
//     fragments of A and W are conflict-free ds_read_b128 from the padded tiles ([.][kp + 4]), one k-group ahead.
//     Two forms that did NOT work: fragment-shaped A loads straight from global memory (16 bytes out of 32 different lines per
//     instruction, 13 times over a 51 KB working set that does not fit the 32 KB L1): 1.70 ms, no better than the tiled kernel;
//     the whole W panel as B fragments in registers (208 per lane): the compiler parks them in AGPRs and pays a v_accvgpr_read
//     per MFMA operand, and spills the staging registers to scratch;
//   * columns behind k inside the last k-group are zero in BOTH LDS tiles (zeroed once), so no row ever reads another row's data
//     (unlike gemm.hip's k_pad_plan) and the result has the k order of the tiled kernels' fragments: bit-identical to them.
// Epilogue: * ep_scale[n] + ep_shift[n], ReLU (bias / eval BatchNorm fold), stored from the accumulators.
#include <cstdlib>

#include "glnn_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kSkMaxKg = 16;             // k-groups of 8: k <= 128
constexpr int kSkCols = 128;             // output columns per workgroup (4 MFMA column blocks per wave)

struct SkinnyArgs {
  const float* a; int64_t lda; int64_t m; int k;
  const float* w; int64_t ldw; int n;
  const float* ep_scale; const float* ep_shift; int relu;
  float* c; int64_t ldc;
  int row_tiles;                          // ceil(m / 256)
  int lds_stride;                         // floats per LDS row: >= k, % 8 == 4
};

template <int KG>      // k-groups of 8 (compile time: the staging array lives in registers)
__global__ __launch_bounds__(512) void gemm_skinny_kernel(const SkinnyArgs g) {
  // LDS: the W panel [128][S] + per wave ONE private [32][S] tile of A; S = g.lds_stride: the smallest S >= k with S % 8 == 4
  // (36 * row mod 64 banks: conflict-free ds_read_b128) -- k = 100 needs no padding at all, and 8 waves (2 per SIMD) fit 160 KB
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int LDW = g.lds_stride;
  const int TILE = 32 * LDW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, kk = lane >> 5;
  const int n0 = blockIdx.y * kSkCols;
  const int kq = g.k >> 2;                                   // float4 per row
  float* w_lds = lds;
  float* my = lds + kSkCols * LDW + wave * TILE;
  // ---- the weight panel: rows n0 .. n0+127 of W, k padded with zeros to KG*8 (+4 pad floats, never read) ----
  const int sq = LDW >> 2;
  for (int i = tid; i < kSkCols * sq; i += 512) {
    const int r = i / sq, c4 = (i - r * sq) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n0 + r < g.n && c4 < g.k) v = *reinterpret_cast<const float4*>(g.w + (int64_t)(n0 + r) * g.ldw + c4);
    *reinterpret_cast<float4*>(w_lds + r * LDW + c4) = v;
  }
  // the pad columns k .. S-1 of the A tile are never written by the loads below: zero them once
  for (int i = lane; i < 32 * (LDW - g.k); i += 64) {
    const int r = i / (LDW - g.k), c = g.k + i % (LDW - g.k);
    my[r * LDW + c] = 0.f;
  }
  __syncthreads();
  // the last k-group may reach behind S (k = 100, S = 100: its upper half, k = 100..103): those A fragments are ZERO registers (the
  // W fragments there are finite LDS contents -- the next panel row -- so the products vanish exactly)
  const bool tail_zero = (KG - 1) * 8 + kk * 4 >= LDW;
  // A sub-tile of this wave: 32 rows x k floats, loaded as lane-linear float4 pieces (coalesced: with lda == k one contiguous block).
  // 32 * kq <= 64 * KG pieces; the pieces behind the tile re-load its last one (unconditional loads)
#define GLNN_SK_LOAD(ROW0_)                                                                    \
  _Pragma("unroll") for (int i = 0; i < KG; ++i) {                                             \
    int p_ = lane + 64 * i;                                                                    \
    if (p_ > 32 * kq - 1) p_ = 32 * kq - 1;                                                    \
    const int r_ = p_ / kq, c4_ = p_ - r_ * kq;                                                \
    int64_t row_ = (ROW0_) + r_;                                                               \
    if (row_ > g.m - 1) row_ = g.m - 1; /* rows behind m: clamped (their outputs are not stored) */ \
    stg[i] = *reinterpret_cast<const float4*>(g.a + row_ * g.lda + c4_ * 4);                   \
  }
#define GLNN_SK_STAGE()                                                                        \
  _Pragma("unroll") for (int i = 0; i < KG; ++i) {                                             \
    const int p_ = lane + 64 * i;                                                              \
    if (p_ < 32 * kq) {                                                                        \
      const int r_ = p_ / kq, c4_ = p_ - r_ * kq;                                              \
      *reinterpret_cast<float4*>(my + r_ * LDW + c4_ * 4) = stg[i];                            \
    }                                                                                          \
  }
  float es[4], eh[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = n0 + j * 32 + li;
    es[j] = (g.ep_scale && col < g.n) ? g.ep_scale[col] : 1.f;
    eh[j] = (g.ep_shift && col < g.n) ? g.ep_shift[col] : 0.f;
  }
  float4 stg[KG];
  int64_t tile = blockIdx.x;
  if (tile < g.row_tiles) {
    GLNN_SK_LOAD(tile * 256 + wave * 32)
    GLNN_SK_STAGE()
  }
  const float* ap = my + li * LDW + kk * 4;
  const float* bp = w_lds + li * LDW + kk * 4;
#pragma unroll 1
  for (; tile < g.row_tiles; tile += gridDim.x) {
    asm volatile("" ::: "memory");            // the LDS tile changes every iteration: keep the fragment reads inside the loop
    const int64_t row0 = tile * 256 + wave * 32;
    const int64_t ntile = tile + gridDim.x;
    { const int64_t lt = ntile < g.row_tiles ? ntile : tile; GLNN_SK_LOAD(lt * 256 + wave * 32) }   // in flight under this tile's MFMAs
                                                                                 // (behind the last tile: a harmless re-load)
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float4 av[2], bq[2][4];
    av[0] = *reinterpret_cast<const float4*>(ap);
#pragma unroll
    for (int j = 0; j < 4; ++j) bq[0][j] = *reinterpret_cast<const float4*>(bp + j * 32 * LDW);
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
      if (kg + 1 < KG) {
        av[(kg + 1) & 1] = (kg + 1 == KG - 1 && tail_zero) ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(ap + (kg + 1) * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) bq[(kg + 1) & 1][j] = *reinterpret_cast<const float4*>(bp + j * 32 * LDW + (kg + 1) * 8);
      }
      const float4 a4 = av[kg & 1];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, bq[kg & 1][j].x, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, bq[kg & 1][j].y, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, bq[kg & 1][j].z, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, bq[kg & 1][j].w, acc[j], 0, 0, 0);
    }
    GLNN_SK_STAGE()                                  // this wave's reads of the tile are done (LDS operations of a wave are ordered)
    // ---- epilogue: C/D map of the 32 x 32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) ----
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = n0 + j * 32 + li;
      if (col < g.n) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
          float v = fmaf(acc[j][r], es[j], eh[j]);
          if (g.relu) v = fmaxf(v, 0.f);
          if (row < g.m) g.c[row * g.ldc + col] = v;
        }
      }
    }
  }
}

#undef GLNN_SK_LOAD
#undef GLNN_SK_STAGE

template <int KG>
int launch_skinny(const SkinnyArgs& g, hipStream_t st) {
  const size_t smem = sizeof(float) * (kSkCols + 8 * 32) * g.lds_stride;
  static int configured = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_skinny_kernel<KG>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              160 * 1024) == hipSuccess ? 0 : -1;
  if (configured != 0) return glnn::fail(GLNN_ERR_HIP, "glnn_gemm_f32(skinny): cannot raise the dynamic LDS limit");
  const int panels = (g.n + kSkCols - 1) / kSkCols;
  const char* ge = getenv("GLNN_SKINNY_WGS");
  int gx = (ge ? atoi(ge) : 256) / panels;  // one workgroup per CU in total (the kernel holds the whole W panel in registers: one wave
                                            // per SIMD); every workgroup keeps its panel for all its row tiles
  if (gx < 1) gx = 1;
  if (gx > g.row_tiles) gx = g.row_tiles;
  hipLaunchKernelGGL(gemm_skinny_kernel<KG>, dim3(gx, panels), dim3(512), smem, st, g);
  return glnn::check_launch("glnn_gemm_f32(skinny)");
}

}  // namespace

namespace glnn {

// Eligibility is decided by the caller (glnn_gemm_f32): NT form, plain A with 16-byte-aligned rows, 36 <= k <= 128, k % 4 == 0.
int gemm_skinny_nt(const float* a, int64_t lda, int64_t m, int k, const float* w, int64_t ldw, int n, const float* ep_scale,
                   const float* ep_shift, int relu, float* c, int64_t ldc, void* stream) {
  SkinnyArgs g;
  g.a = a; g.lda = lda; g.m = m; g.k = k; g.w = w; g.ldw = ldw; g.n = n; g.ep_scale = ep_scale; g.ep_shift = ep_shift; g.relu = relu;
  g.c = c; g.ldc = ldc; g.row_tiles = (int)((m + 255) / 256);
  g.lds_stride = (k % 8 == 4) ? k : ((k + 7) / 8 * 8 + 4);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int kg = (k + 7) / 8;
  if (kg <= 5) return launch_skinny<5>(g, st);
  if (kg <= 8) return launch_skinny<8>(g, st);
  if (kg <= 13) return launch_skinny<13>(g, st);
  return launch_skinny<16>(g, st);
}

}  // namespace glnn
