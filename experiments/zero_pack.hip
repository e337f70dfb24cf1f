// EXPERIMENT (not part of libglnn_hip.so), round 5: the round-3 lossless compressed gather for post-ReLU activations,
// D = 256, with the header IN the row's own slot instead of a side array (the side array cost a sixth, half-used line per edge):
//   row slot of 288 floats (1152 B = 9 lines, line-aligned): floats 0..15 = 8 x {mask of 32 columns, non-zeros in the words before},
//   floats 16.. = the row's non-zeros; a row of 128 non-zeros touches 64 + 512 = 576 B = 5 lines instead of the dense 8.
//   (row format of the side-array form: meta[row][8] = {mask (32 columns), prefix}; packed[row][0..nnz) values)
//   gather: lane = 4 columns; meta word = lane/8; position = prefix + popc(mask below the lane's nibble); one (unaligned)
//           16-byte load of the next 4 packed values; expand by the nibble.
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o experiments/libzeropack.so experiments/zero_pack.hip
#include <hip/hip_runtime.h>
#include <cstdint>

namespace {
constexpr int D = 256;
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // 16 bytes at a 4-byte-aligned address
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

__global__ __launch_bounds__(256) void compress_kernel(const float* __restrict__ x, int64_t ldx, int64_t n, uint2* __restrict__ meta,
                                                        float* __restrict__ packed, int64_t ldp) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const float4 v = *reinterpret_cast<const float4*>(x + row * ldx + 4 * lane);
  const unsigned nib = (v.x != 0.f ? 1u : 0u) | (v.y != 0.f ? 2u : 0u) | (v.z != 0.f ? 4u : 0u) | (v.w != 0.f ? 8u : 0u);
  const int cnt = __popc(nib);
  int incl = cnt;                                  // inclusive scan over the 64 lanes
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  const int pos = incl - cnt;
  unsigned word = nib << (4 * (lane & 7));
  word |= __shfl_xor(word, 1); word |= __shfl_xor(word, 2); word |= __shfl_xor(word, 4);
  if ((lane & 7) == 0) {
    if (meta) meta[row * 8 + (lane >> 3)] = make_uint2(word, (unsigned)pos);
    else reinterpret_cast<uint2*>(packed + row * ldp)[lane >> 3] = make_uint2(word, (unsigned)pos);
  }
  float* p = packed + row * ldp + pos + (meta ? 0 : 16);
  int k = 0;
  if (nib & 1u) p[k++] = v.x;
  if (nib & 2u) p[k++] = v.y;
  if (nib & 4u) p[k++] = v.z;
  if (nib & 8u) p[k++] = v.w;
}

template <int U, bool SPARSE>
__global__ __launch_bounds__(512) void gather_kernel(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices, int64_t n_dst,
                                                      const float* __restrict__ x, int64_t ldx, const uint2* __restrict__ meta,
                                                      float* __restrict__ out, int64_t ldo) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  __shared__ int s_ticket;
  if (threadIdx.x == 0) s_ticket = 0;
  __syncthreads();
  const int64_t row_base = (int64_t)blockIdx.x * 128;
  const int w = lane >> 3, sh = 4 * (lane & 7);
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
      const int64_t rem = e1 - base;
      const int cnt = rem < 64 ? (int)rem : 64;
      const int my_idx = lane < cnt ? __builtin_nontemporal_load(indices + base + lane) : 0;
      for (int j = 0; j < cnt; j += U) {
        float4 q[U];
        unsigned nib[U];
        if (SPARSE) {
          uint2 m[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int src = __builtin_amdgcn_readlane(my_idx, (j + u) & 63);
            m[u] = (j + u >= cnt) ? make_uint2(0u, 0u) : (meta ? meta[(int64_t)src * 8 + w] : reinterpret_cast<const uint2*>(x + (int64_t)src * ldx)[w]);
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int src = __builtin_amdgcn_readlane(my_idx, (j + u) & 63);
            nib[u] = (m[u].x >> sh) & 0xFu;
            const int pos = (int)m[u].y + __popc(m[u].x & ((1u << sh) - 1u));
            const float* p = x + (int64_t)src * ldx + pos + (meta ? 0 : 16);
            // 16 bytes at a 4-byte-aligned address: the next four packed values (only popc(nib) of them are this lane's)
            if (j + u < cnt && nib[u]) {
              const f4u t = *reinterpret_cast<const f4u*>(p);
              q[u] = make_float4(t.x, t.y, t.z, t.w);
            } else {
              q[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const unsigned b = nib[u];
            const float4 t = q[u];
            const float y0 = t.x;
            const float y1 = (b & 1u) ? t.y : t.x;
            const int i2 = __popc(b & 3u);
            const float y2 = i2 == 0 ? t.x : (i2 == 1 ? t.y : t.z);
            const int i3 = __popc(b & 7u);
            const float y3 = i3 == 0 ? t.x : (i3 == 1 ? t.y : (i3 == 2 ? t.z : t.w));
            acc.x += (b & 1u) ? y0 : 0.f;
            acc.y += (b & 2u) ? y1 : 0.f;
            acc.z += (b & 4u) ? y2 : 0.f;
            acc.w += (b & 8u) ? y3 : 0.f;
          }
        } else {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int src = __builtin_amdgcn_readlane(my_idx, (j + u) & 63);
            q[u] = (j + u < cnt) ? *reinterpret_cast<const float4*>(x + (int64_t)src * ldx + 4 * lane) : make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (int u = 0; u < U; ++u) acc = add4(acc, q[u]);
        }
      }
    }
    *reinterpret_cast<float4*>(out + v * ldo + 4 * lane) = acc;
  }
  (void)wave;
}
}  // namespace

extern "C" __attribute__((visibility("default"))) int sp_compress(const float* x, int64_t ldx, int64_t n, void* meta, float* packed, int64_t ldp, void* st) {
  hipLaunchKernelGGL(compress_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)st, x, ldx, n, (uint2*)meta, packed, ldp);
  return (int)hipGetLastError();
}
extern "C" __attribute__((visibility("default"))) int sp_gather(const int64_t* indptr, const int32_t* indices, int64_t n_dst, const float* x, int64_t ldx,
                                                                const void* meta, float* out, int64_t ldo, int sparse, void* st) {
  const dim3 grid((unsigned)((n_dst + 127) / 128));
  if (sparse == 2) hipLaunchKernelGGL((gather_kernel<12, true>), grid, dim3(512), 0, (hipStream_t)st, indptr, indices, n_dst, x, ldx, (const uint2*)meta, out, ldo);
  else if (sparse == 3) hipLaunchKernelGGL((gather_kernel<16, true>), grid, dim3(512), 0, (hipStream_t)st, indptr, indices, n_dst, x, ldx, (const uint2*)meta, out, ldo);
  else if (sparse) hipLaunchKernelGGL((gather_kernel<8, true>), grid, dim3(512), 0, (hipStream_t)st, indptr, indices, n_dst, x, ldx, (const uint2*)meta, out, ldo);
  else hipLaunchKernelGGL((gather_kernel<8, false>), grid, dim3(512), 0, (hipStream_t)st, indptr, indices, n_dst, x, ldx, (const uint2*)meta, out, ldo);
  return (int)hipGetLastError();
}
