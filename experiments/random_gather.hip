// EXPERIMENT (round 6; not part of libglnn_hip.so): what does the part deliver for RANDOM row gathers with no cache
// hits -- the DRAM-side ceiling of the aggregation at XL size, where layer 3 gathers 188-byte rows (2 lines) out of a 19 GB buffer and
// runs at 3.8 TB/s of algorithmic bytes = 5.15 TB/s of 128-byte lines, against 7.5 TB/s of lines for the same kernel on the products
// graph (whose 0.47 GB source matrix half lives in the Infinity Cache)?
//
// The binary gathers rows of `d` floats (pitch ld) from buffers of 1 / 19 / 100 GB with the aggregation's own access shape -- LPR lanes x
// one float4 per row, U = 8 rows in flight per lane group, plain global_load_dwordx4, 512-thread workgroups, the grid the aggregation
// uses -- for
//     d = 47 / ld 48   (192-byte rows: 2 lines)         the XL layer-3 shape
//     d = 100 / ld 100 (400-byte rows: 4 lines mostly)  the products layer-1 shape
//     d = 64 / ld 64   (256-byte aligned rows: exactly 2 lines)   the "512-byte (4-line)" companion is d = 128 / ld 128
// over an edge list of degree-20 destination rows (XL: 2 B edges / 100 M rows), in three orders:
//     random       every edge's source row uniform in the buffer
//     sorted       the same lists with each destination's 20 sources sorted by id (what a CSR with sorted neighbour lists gives)
//     median       ... and the destination rows dealt to the waves in the order of their median source id (neighbouring waves then
//                  walk neighbouring parts of the buffer)
// and prints rows/s, algorithmic TB/s and TB/s of touched 128-byte lines.  If the no-hit ceiling is ~5.2 TB/s of lines, XL layer 3 is AT
// the part's random-access rate and the item closes; if it is materially higher, the kernel has headroom at XL.
//   hipcc --offload-arch=gfx950 -O3 experiments/random_gather.hip -o experiments/random_gather
//   experiments/random_gather [GB ...]        (default: 1 19 100)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int LPR>
__global__ __launch_bounds__(512) void gather_kernel(const int32_t* __restrict__ idx, int64_t n_edges, const float* __restrict__ x, int64_t ld, int d,
                                                      float* __restrict__ out) {
  constexpr int G = 64 / LPR, U = 8;
  const int lane = threadIdx.x & 63, g = lane / LPR, c4 = (lane % LPR) * 4;
  const bool on = c4 < d;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // a wave owns CONSECUTIVE groups of G * U edges (the lists of whole destination rows when 20 | G U is arranged by the host: the order of
  // the edge array is the order the waves walk)
  for (int64_t e0 = wave * (G * U); e0 + G * U <= n_edges; e0 += n_waves * (G * U)) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = idx[e0 + u * G + g];
      const float* p = x + row * ld + c4;
      v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (on) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[u]) : "v"(p) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < U; ++u) { asm volatile("" : "+v"(v[u])); acc += v[u]; }
  }
  if (on) out[(wave * 64 + lane) % (1 << 20)] = acc.x + acc.y + acc.z + acc.w;
}

__global__ void fill_kernel(float* x, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] = 1.0f;
}

template <int LPR>
static double run(const int32_t* idx, int64_t n_edges, const float* x, int64_t ld, int d, float* out) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((gather_kernel<LPR>), dim3(4096), dim3(512), 0, 0, idx, n_edges, x, ld, d, out);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (rep > 0 && ms < best) best = ms;
  }
  return best;
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline uint64_t rng() { uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

int main(int argc, char** argv) {
  std::vector<double> gbs;
  for (int i = 1; i < argc; ++i) gbs.push_back(atof(argv[i]));
  if (gbs.empty()) gbs = {1, 19, 100};
  const int64_t deg = 20, n_rows_dst = 2ll << 20, n_edges = n_rows_dst * deg;      // 42 M edges: 10-22 GB of gathered lines per launch (host-side generation of more costs minutes of box time)
  struct Shape { int d; int64_t ld; int lpr; const char* what; };
  const Shape shapes[] = {{47, 48, 16, "d=47 ld=48 (192 B rows, 2 lines: XL layer 3)"}, {64, 64, 16, "d=64 ld=64 (256 B aligned rows, 2 lines)"},
                          {100, 100, 32, "d=100 ld=100 (400 B rows, 4-5 lines: products layer 1)"}, {128, 128, 32, "d=128 ld=128 (512 B aligned rows, 4 lines)"}};
  int32_t* d_idx; float* d_out;
  CK(hipMalloc(&d_idx, n_edges * sizeof(int32_t)));
  CK(hipMalloc(&d_out, (1 << 20) * sizeof(float)));
  std::vector<int32_t> h(n_edges);
  for (double gb : gbs) {
    const int64_t bytes = (int64_t)(gb * 1e9);
    float* x;
    if (hipMalloc(&x, bytes) != hipSuccess) { printf("buffer of %.0f GB: allocation failed, skipped\n", gb); continue; }
    hipLaunchKernelGGL(fill_kernel, dim3(8192), dim3(256), 0, 0, x, bytes / 4);
    CK(hipDeviceSynchronize());
    for (const Shape& s : shapes) {
      const int64_t n_src = bytes / (s.ld * 4);
      if (n_src > 0x7fffffffll) { printf("%.0f GB / %s: more than 2^31 rows, skipped\n", gb, s.what); continue; }
      // average 128-byte lines a row touches (rows of ld * 4 bytes starting at multiples of ld * 4, d * 4 bytes used)
      double lines = 0;
      for (int64_t r = 0; r < 4096; ++r) { const int64_t b0 = r * s.ld * 4, b1 = b0 + s.d * 4 - 1; lines += (double)(b1 / 128 - b0 / 128 + 1); }
      lines /= 4096;
      for (int order = 0; order < 3; ++order) {
        rng_state = 12345;
        for (int64_t e = 0; e < n_edges; ++e) h[e] = (int32_t)(rng() % (uint64_t)n_src);
        if (order >= 1)
          for (int64_t r = 0; r < n_rows_dst; ++r) std::sort(h.begin() + r * deg, h.begin() + (r + 1) * deg);
        if (order == 2) {      // destination rows in the order of their median source
          std::vector<int32_t> perm(n_rows_dst);
          std::iota(perm.begin(), perm.end(), 0);
          std::sort(perm.begin(), perm.end(), [&](int32_t a, int32_t b) { return h[(int64_t)a * deg + deg / 2] < h[(int64_t)b * deg + deg / 2]; });
          std::vector<int32_t> h2(n_edges);
          for (int64_t r = 0; r < n_rows_dst; ++r) std::copy(h.begin() + (int64_t)perm[r] * deg, h.begin() + ((int64_t)perm[r] + 1) * deg, h2.begin() + r * deg);
          h.swap(h2);
        }
        CK(hipMemcpy(d_idx, h.data(), n_edges * sizeof(int32_t), hipMemcpyHostToDevice));
        const double ms = s.lpr == 16 ? run<16>(d_idx, n_edges, x, s.ld, s.d, d_out) : run<32>(d_idx, n_edges, x, s.ld, s.d, d_out);
        printf("%5.0f GB  %-58s %-7s %8.2f ms  %6.2f G rows/s  %5.2f TB/s algorithmic (d*4+4 per edge)  %5.2f TB/s of 128-byte lines (%.2f lines/row)\n", gb, s.what,
               order == 0 ? "random" : order == 1 ? "sorted" : "median", ms, n_edges / ms / 1e6, n_edges * (s.d * 4.0 + 4.0) / ms / 1e9,
               n_edges * lines * 128.0 / ms / 1e9, lines);
        fflush(stdout);
      }
    }
    CK(hipFree(x));
  }
  return 0;
}
