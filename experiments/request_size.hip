// EXPERIMENT (round 4; not part of libglnn_hip.so): can the tail of a gathered feature row be fetched from the fabric
// with a request smaller than a 128-byte line?  The aggregation's traffic is 1.26-1.34 x its algorithmic bytes because rows of
// 188 B (D = 47) and 400 B (D = 100) are fetched as whole 128-byte lines (profiles/pmc_l2_r03.json: RDREQ_32B = RDREQ_64B = 0).
// This binary gathers random rows of a matrix far larger than the 256 MB Infinity Cache with the aggregation's access shape -- one
// float4 per lane, LPR lanes per row, U rows in flight per lane group -- through every cache-policy form of global_load_dwordx4
// gfx950 assembles (plain | nt | sc0 | sc1 | sc0 sc1 | nt sc0 sc1), and, for the row TAIL only (the lanes behind the row's last full
// 128-byte line), with the policy forms and with narrower per-lane widths (dwordx2, dword).  Run under
//   rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
// (scripts/request_size.sh) to read the request sizes per variant; stdout carries the times.
//   hipcc --offload-arch=gfx950 -O3 experiments/request_size.hip -o experiments/request_size
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define LD4(SUFFIX) asm volatile("global_load_dwordx4 %0, %1, off" SUFFIX : "=v"(v) : "v"(p) : "memory")

template <int POLICY>
__device__ __forceinline__ f32x4 ld4p(const float* p) {
  f32x4 v;
  if (POLICY == 0) LD4("");
  if (POLICY == 1) LD4(" nt");
  if (POLICY == 2) LD4(" sc0");
  if (POLICY == 3) LD4(" sc1");
  if (POLICY == 4) LD4(" sc0 sc1");
  if (POLICY == 5) LD4(" nt sc0 sc1");
  return v;
}
// narrow forms for the tail lanes: the 16 bytes of a lane as two dwordx2 / four dword loads (plain policy)
__device__ __forceinline__ f32x4 ld4_x2(const float* p) {
  f32x2 a, b;
  asm volatile("global_load_dwordx2 %0, %2, off\n\tglobal_load_dwordx2 %1, %2, off offset:8" : "=&v"(a), "=&v"(b) : "v"(p) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b));
  return f32x4{a.x, a.y, b.x, b.y};
}
__device__ __forceinline__ f32x4 ld4_x1(const float* p) {
  float a, b, c, d;
  asm volatile("global_load_dword %0, %4, off\n\tglobal_load_dword %1, %4, off offset:4\n\tglobal_load_dword %2, %4, off offset:8\n\t"
               "global_load_dword %3, %4, off offset:12" : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  return f32x4{a, b, c, d};
}

// rows of `d` floats (row stride ld), LPR lanes x float4 per row; BODY = policy of the lanes inside full lines, TAIL = form of the
// lanes whose float4 lies in the row's last, partial 128-byte line (TAIL < 6: policy; 6: dwordx2 x 2; 7: dword x 4)
template <int LPR, int BODY, int TAIL>
__global__ __launch_bounds__(512) void gather_kernel(const int32_t* __restrict__ idx, int64_t n_edges, const float* __restrict__ x, int ld, int d,
                                                      float* __restrict__ out) {
  constexpr int G = 64 / LPR, U = 8;
  const int lane = threadIdx.x & 63, g = lane / LPR, c4 = (lane % LPR) * 4;
  const bool on = c4 < d;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int64_t e0 = wave * (G * U); e0 + G * U <= n_edges; e0 += n_waves * (G * U)) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = idx[e0 + u * G + g];
      const float* p = x + row * ld + c4;
      // is this lane's float4 inside the row's last partial line?  (row start byte = row*ld*4; the row ends at +d*4)
      const uint64_t b0 = (uint64_t)(row * ld + c4) * 4, bend = (uint64_t)(row * ld + d) * 4;
      const bool tail = (b0 / 128) == ((bend - 1) / 128) && (bend % 128) != 0;
      v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (on) {
        if (BODY == TAIL || !tail) v[u] = ld4p<BODY>(p);
        else if (TAIL < 6) v[u] = ld4p<TAIL>(p);
        else if (TAIL == 6) v[u] = ld4_x2(p);
        else v[u] = ld4_x1(p);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < U; ++u) { asm volatile("" : "+v"(v[u])); acc += v[u]; }
  }
  if (on) out[(wave * 64 + lane) % (1 << 20)] = acc.x + acc.y + acc.z + acc.w;
}

// SPLIT LAYOUT probe: a D-column row stored as a 128-byte-aligned MAIN part (`dm` columns, pitch ldm: whole lines only) + a small
// TAIL part (d - dm columns, pitch ldt) in a second, compact array that can stay in the 256 MB Infinity Cache: does a line served by the
// Infinity Cache cost as much as one from HBM?  MODE 0 = both parts, 1 = main part only (upper bound), 2 = tail part only.
template <int LPR, int MODE>
__global__ __launch_bounds__(512) void gather_split_kernel(const int32_t* __restrict__ idx, int64_t n_edges, const float* __restrict__ xm, int ldm, int dm,
                                                            const float* __restrict__ xt, int ldt, int d, float* __restrict__ out) {
  constexpr int G = 64 / LPR, U = 8;
  const int lane = threadIdx.x & 63, g = lane / LPR, c4 = (lane % LPR) * 4;
  const bool in_main = c4 < dm, on = c4 < d && (MODE == 0 || (MODE == 1) == in_main);
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int64_t e0 = wave * (G * U); e0 + G * U <= n_edges; e0 += n_waves * (G * U)) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = idx[e0 + u * G + g];
      const float* p = in_main ? xm + row * ldm + c4 : xt + row * ldt + (c4 - dm);
      v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (on) v[u] = ld4p<0>(p);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < U; ++u) { asm volatile("" : "+v"(v[u])); acc += v[u]; }
  }
  if (on) out[(wave * 64 + lane) % (1 << 20)] = acc.x + acc.y + acc.z + acc.w;
}

template <int LPR, int MODE>
void run_split(const char* what, const int32_t* idx, int64_t n_edges, const float* xm, int ldm, int dm, const float* xt, int ldt, int d, float* out) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL((gather_split_kernel<LPR, MODE>), dim3(4096), dim3(512), 0, 0, idx, n_edges, xm, ldm, dm, xt, ldt, d, out);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  printf("split d=%3d main %3d cols (pitch %3d) + tail (pitch %2d) mode=%d  %-30s %8.3f ms  %6.2f G rows/s\n", d, dm, ldm, ldt, MODE, what, best,
         (double)n_edges / best / 1e6);
  fflush(stdout);
}

template <int LPR, int BODY, int TAIL>
float run(const char* what, const int32_t* idx, int64_t n_edges, const float* x, int ld, int d, float* out) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL((gather_kernel<LPR, BODY, TAIL>), dim3(4096), dim3(512), 0, 0, idx, n_edges, x, ld, d, out);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  const double bytes = (double)n_edges * d * 4;
  printf("d=%3d LPR=%2d body=%d tail=%d  %-34s %8.3f ms  %7.1f GB/s of row bytes\n", d, LPR, BODY, TAIL, what, best, bytes / best / 1e6);
  fflush(stdout);
  return best;
}

int main() {
  const int64_t n_rows = 20'000'000, n_edges = 64'000'000;
  const int64_t floats = n_rows * 100;                         // 8 GB: both layouts (ld = 48 / ld = 100) fit
  float* x; int32_t* idx; float* out;
  if (hipMalloc(&x, floats * 4) != hipSuccess || hipMalloc(&idx, n_edges * 4) != hipSuccess || hipMalloc(&out, 4 << 20) != hipSuccess) return 1;
  (void)hipMemset(x, 0, floats * 4);
  std::vector<int32_t> h(n_edges);
  uint64_t s = 88172645463325252ull;
  for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (int32_t)(s % (uint64_t)n_rows); }
  (void)hipMemcpy(idx, h.data(), n_edges * 4, hipMemcpyHostToDevice);
  // D = 47: rows of 188 B in a 192-byte pitch (12 float4 lanes of 16)
  run<16, 0, 0>("plain", idx, n_edges, x, 48, 47, out);
  run<16, 1, 1>("nt", idx, n_edges, x, 48, 47, out);
  run<16, 2, 2>("sc0", idx, n_edges, x, 48, 47, out);
  run<16, 3, 3>("sc1", idx, n_edges, x, 48, 47, out);
  run<16, 4, 4>("sc0 sc1", idx, n_edges, x, 48, 47, out);
  run<16, 5, 5>("nt sc0 sc1", idx, n_edges, x, 48, 47, out);
  run<16, 0, 1>("plain, tail nt", idx, n_edges, x, 48, 47, out);
  run<16, 0, 4>("plain, tail sc0 sc1", idx, n_edges, x, 48, 47, out);
  run<16, 0, 5>("plain, tail nt sc0 sc1", idx, n_edges, x, 48, 47, out);
  run<16, 0, 6>("plain, tail 2 x dwordx2", idx, n_edges, x, 48, 47, out);
  run<16, 0, 7>("plain, tail 4 x dword", idx, n_edges, x, 48, 47, out);
  // D = 100: rows of 400 B, pitch 400 (25 float4 lanes of 32)
  run<32, 0, 0>("plain", idx, n_edges, x, 100, 100, out);
  run<32, 1, 1>("nt", idx, n_edges, x, 100, 100, out);
  run<32, 4, 4>("sc0 sc1", idx, n_edges, x, 100, 100, out);
  run<32, 5, 5>("nt sc0 sc1", idx, n_edges, x, 100, 100, out);
  run<32, 0, 5>("plain, tail nt sc0 sc1", idx, n_edges, x, 100, 100, out);
  run<32, 0, 7>("plain, tail 4 x dword", idx, n_edges, x, 100, 100, out);
  // ---- split layout, products-sized row range (2.45 M rows: the tail arrays are 39 MB / 157 MB) ----
  {
    const int64_t n_small = 2'449'029;
    std::vector<int32_t> h2(n_edges);
    for (auto& v : h2) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (int32_t)(s % (uint64_t)n_small); }
    (void)hipMemcpy(idx, h2.data(), n_edges * 4, hipMemcpyHostToDevice);
    float* xt = x + n_small * 128;                           // behind the main part (pitch <= 128)
    run<32, 0, 0>("plain pitch 400, 2.45 M rows", idx, n_edges, x, 100, 100, out);
    run_split<32, 0>("96 + 4", idx, n_edges, x, 96, 96, xt, 4, 100, out);
    run_split<32, 1>("96 only (bound)", idx, n_edges, x, 96, 96, xt, 4, 100, out);
    run_split<32, 2>("tail 4 only", idx, n_edges, x, 96, 96, xt, 4, 100, out);
    run<16, 0, 0>("plain pitch 192, 2.45 M rows", idx, n_edges, x, 48, 47, out);
    run_split<16, 0>("32 + 16", idx, n_edges, x, 32, 32, xt, 16, 47, out);
    run_split<16, 1>("32 only (bound)", idx, n_edges, x, 32, 32, xt, 16, 47, out);
    run_split<16, 2>("tail 16 only", idx, n_edges, x, 32, 32, xt, 16, 47, out);
  }
  return 0;
}
