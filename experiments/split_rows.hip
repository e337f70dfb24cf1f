// EXPERIMENT: D = 100 gather with the row split at the last 128-byte multiple: main[N][96] (384-byte rows = 3 whole lines)
// + rem[N][4] (16 bytes per node, 39 MB for products: meant to stay cache-resident) vs the plain [N][100] layout whose
// 400-byte rows straddle 4.1 lines on average.
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o experiments/libsplit.so experiments/split_rows.hip
#include <hip/hip_runtime.h>
#include <cstdint>
namespace {
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
// 32 lanes per row (25 used), two edges per wave step, U steps in flight
template <int U, bool SPLIT>
__global__ __launch_bounds__(512) void gather100(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices, int64_t n_dst,
                                                 const float* __restrict__ x, int64_t ldx, const float* __restrict__ rem,
                                                 float* __restrict__ out, int64_t ldo) {
  const int lane = threadIdx.x & 63;
  const int c = lane & 31, grp = lane >> 5;
  const bool col_ok = c < 25;
  __shared__ int s_ticket;
  if (threadIdx.x == 0) s_ticket = 0;
  __syncthreads();
  const int64_t row_base = (int64_t)blockIdx.x * 128;
  while (true) {
    int lr = 0;
    if (lane == 0) lr = atomicAdd(&s_ticket, 1);
    lr = __builtin_amdgcn_readfirstlane(lr);
    if (lr >= 128) break;
    const int64_t v = row_base + lr;
    if (v >= n_dst) break;
    const int64_t e0 = indptr[v], e1 = indptr[v + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t base = e0; base < e1; base += 64) {
      const int64_t remn = e1 - base;
      const int cnt = remn < 64 ? (int)remn : 64;
      const int my_idx = lane < cnt ? __builtin_nontemporal_load(indices + base + lane) : 0;
      for (int j = 0; j < cnt; j += 2 * U) {
        float4 q[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int ei = j + 2 * u + grp;
          const int src = __shfl(my_idx, ei & 63);
          const float* p;
          if (SPLIT) p = (c < 24) ? x + (int64_t)src * ldx + 4 * c : rem + (int64_t)src * 4;
          else p = x + (int64_t)src * ldx + 4 * c;
          q[u] = (ei < cnt && col_ok) ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc = add4(acc, q[u]);
      }
    }
    acc.x += __shfl_xor(acc.x, 32); acc.y += __shfl_xor(acc.y, 32); acc.z += __shfl_xor(acc.z, 32); acc.w += __shfl_xor(acc.w, 32);
    if (lane < 25) *reinterpret_cast<float4*>(out + v * ldo + 4 * lane) = acc;
  }
}
}  // namespace
extern "C" __attribute__((visibility("default"))) int sp_gather100(const int64_t* indptr, const int32_t* indices, int64_t n_dst, const float* x, int64_t ldx,
                                                                   const float* rem, float* out, int64_t ldo, int split, void* st) {
  const dim3 grid((unsigned)((n_dst + 127) / 128));
  if (split) hipLaunchKernelGGL((gather100<8, true>), grid, dim3(512), 0, (hipStream_t)st, indptr, indices, n_dst, x, ldx, rem, out, ldo);
  else hipLaunchKernelGGL((gather100<8, false>), grid, dim3(512), 0, (hipStream_t)st, indptr, indices, n_dst, x, ldx, rem, out, ldo);
  return (int)hipGetLastError();
}
