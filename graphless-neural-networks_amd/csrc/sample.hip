// Uniform neighbour sampling without replacement on the resident CSR (row f-2 of SURVEY.md section 8):
// replaces the CPU-side dgl.dataloading.MultiLayerNeighborSampler the reference drives per batch
// (reference train_and_eval.py:179-190; fan-out "5,10,15" from train.conf.yaml) -- integer, HBM-bound work.
// One thread per seed row: rows with in-degree <= fanout keep all their in-edges (like dgl); otherwise
// Floyd's algorithm draws `fanout` distinct edge positions with a counter-based hash RNG.
#include "glnn_common.h"

namespace {

__device__ __forceinline__ uint32_t rng32(uint32_t seed, uint32_t a, uint32_t b) { return glnn::drop_hash(seed, a, b); }

// F = compile-time bound of the fan-out (8 / 16 / 32 / 64): with every loop over the picked positions fully unrolled the
// `picked` array stays in registers (a runtime-indexed int64[64] lived in scratch memory: 26-75 us per block of a training batch).
template <int F>
__global__ __launch_bounds__(256) void sample_neighbors_kernel(const int64_t* __restrict__ indptr,
                                                               const int32_t* __restrict__ indices,
                                                               const int64_t* __restrict__ seeds, int64_t n_seeds, int fanout,
                                                               uint32_t seed, int32_t* __restrict__ out_src,
                                                               int32_t* __restrict__ out_cnt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_seeds) return;
  const int64_t v = seeds[i];
  const int64_t e0 = indptr[v];
  const int64_t deg = indptr[v + 1] - e0;
  int32_t* dst = out_src + i * fanout;
  if (deg <= fanout) {
    for (int64_t k = 0; k < deg; ++k) dst[k] = indices[e0 + k];
    out_cnt[i] = (int32_t)deg;
    return;
  }
  // Floyd: for j = deg-fanout .. deg-1: t = U[0, j]; pick t unless already picked, then pick j
  int64_t picked[F];
#pragma unroll
  for (int s = 0; s < F; ++s) picked[s] = -1;
  const int64_t base = deg - fanout;
#pragma unroll
  for (int s = 0; s < F; ++s) {
    if (s < fanout) {
      const int64_t j = base + s;
      const uint32_t r = rng32(seed, (uint32_t)i, (uint32_t)s);
      int64_t t = (int64_t)(((uint64_t)r * (uint64_t)(j + 1)) >> 32);
      bool dup = false;
#pragma unroll
      for (int q = 0; q < F; ++q) dup |= (q < s) && (picked[q] == t);
      picked[s] = dup ? j : t;
    }
  }
#pragma unroll
  for (int s = 0; s < F; ++s)
    if (s < fanout) dst[s] = indices[e0 + picked[s]];
  out_cnt[i] = fanout;
}

}  // namespace

extern "C" int glnn_sample_neighbors(const int64_t* indptr, const int32_t* indices, const int64_t* seeds, int64_t n_seeds,
                                     int fanout, uint32_t rng_seed, int32_t* out_src, int32_t* out_cnt, void* stream) {
  GLNN_REQUIRE(indptr && indices && seeds && out_src && out_cnt, "glnn_sample_neighbors: null pointer");
  GLNN_REQUIRE(n_seeds >= 0 && fanout >= 1 && fanout <= 64, "glnn_sample_neighbors: fanout must be in [1,64]");
  if (n_seeds == 0) return GLNN_OK;
  const dim3 grid((unsigned)((n_seeds + 255) / 256));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define GLNN_SAMPLE(F_) hipLaunchKernelGGL((sample_neighbors_kernel<F_>), grid, dim3(256), 0, st, indptr, indices, seeds, n_seeds, fanout, rng_seed, out_src, out_cnt)
  if (fanout <= 8) GLNN_SAMPLE(8);
  else if (fanout <= 16) GLNN_SAMPLE(16);
  else if (fanout <= 32) GLNN_SAMPLE(32);
  else GLNN_SAMPLE(64);
#undef GLNN_SAMPLE
  return glnn::check_launch("glnn_sample_neighbors");
}
