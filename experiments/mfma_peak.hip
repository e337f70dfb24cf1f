// Sustained v_mfma_f32_32x32x2_f32 rate of this box, registers only (no LDS, no global traffic inside the loop): the
// practical ceiling the fp32 GEMMs of the student step are priced against next to the 157 TFLOP/s nominal peak.
//   hipcc --offload-arch=gfx950 -O3 experiments/mfma_peak.hip -o experiments/mfma_peak && experiments/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int ACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float seed) {
  f32x16 acc[ACC];
#pragma unroll
  for (int i = 0; i < ACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = seed + threadIdx.x, b = seed * 0.5f + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < ACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int ACC>
static void run(const char* name, int wgs, int iters, float* out, int repeat_ms) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_loop<ACC>, dim3(wgs), dim3(256), 0, 0, out, iters, 1.0f);
  hipDeviceSynchronize();
  // several back-to-back launches: the first ones run at boost clocks, sustained load settles lower
  for (int rep = 0; rep < 4; ++rep) {
    const int launches = repeat_ms;
    hipEventRecord(e0, 0);
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(mfma_loop<ACC>, dim3(wgs), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)launches * wgs * 4.0 * iters * 8.0 * ACC * 4096.0;
    printf("%-28s wgs %5d  rep %d: %8.3f ms  %7.1f TFLOP/s\n", name, wgs, rep, ms, flops / ms / 1e9);
  }
}

int main() {
  float* out;
  hipMalloc(&out, 4096 * 256 * sizeof(float));
  run<4>("4 acc, 1 wave/SIMD", 256, 4000, out, 20);
  run<4>("4 acc, 2 waves/SIMD", 512, 4000, out, 10);
  run<8>("8 acc, 1 wave/SIMD", 256, 2000, out, 20);
  run<4>("4 acc, 4 waves/SIMD", 1024, 4000, out, 5);
  run<2>("2 acc, 2 waves/SIMD", 512, 8000, out, 10);
  hipFree(out);
  return 0;
}
