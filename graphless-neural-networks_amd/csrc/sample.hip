// Uniform neighbour sampling without replacement on the resident CSR (row f-2 of SURVEY.md section 8):
// replaces the CPU-side dgl.dataloading.MultiLayerNeighborSampler the reference drives per batch
// (reference train_and_eval.py:179-190; fan-out "5,10,15" from train.conf.yaml) -- integer, HBM-bound work.
// One thread per seed row: rows with in-degree <= fanout keep all their in-edges (like dgl); otherwise
// Floyd's algorithm draws `fanout` distinct edge positions with a counter-based hash RNG.
#include "glnn_common.h"

namespace {

__device__ __forceinline__ uint32_t rng32(uint32_t seed, uint32_t a, uint32_t b) { return glnn::drop_hash(seed, a, b); }

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
  int64_t picked[64];
  int np = 0;
  for (int64_t j = deg - fanout; j < deg; ++j) {
    const uint32_t r = rng32(seed, (uint32_t)i, (uint32_t)(j - (deg - fanout)));
    int64_t t = (int64_t)(((uint64_t)r * (uint64_t)(j + 1)) >> 32);
    bool dup = false;
    for (int q = 0; q < np; ++q) dup |= (picked[q] == t);
    if (dup) t = j;
    picked[np++] = t;
  }
  for (int k = 0; k < fanout; ++k) dst[k] = indices[e0 + picked[k]];
  out_cnt[i] = fanout;
}

}  // namespace

extern "C" int glnn_sample_neighbors(const int64_t* indptr, const int32_t* indices, const int64_t* seeds, int64_t n_seeds,
                                     int fanout, uint32_t rng_seed, int32_t* out_src, int32_t* out_cnt, void* stream) {
  GLNN_REQUIRE(indptr && indices && seeds && out_src && out_cnt, "glnn_sample_neighbors: null pointer");
  GLNN_REQUIRE(n_seeds >= 0 && fanout >= 1 && fanout <= 64, "glnn_sample_neighbors: fanout must be in [1,64]");
  if (n_seeds == 0) return GLNN_OK;
  hipLaunchKernelGGL(sample_neighbors_kernel, dim3((unsigned)((n_seeds + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), indptr, indices, seeds, n_seeds, fanout, rng_seed, out_src, out_cnt);
  return glnn::check_launch("glnn_sample_neighbors");
}
