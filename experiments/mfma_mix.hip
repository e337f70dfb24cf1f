// How much do LDS stores / global loads issued between MFMAs cost, per MFMA shape?  Registers-only MFMA loop (as mfma_peak.hip)
// plus, every STEP cycles' worth of MFMAs, one ds_write_b128 and/or one global_load_dwordx4 (same bytes per flop for both shapes:
// the staging rate of a 128x128x32 GEMM tile = 8 + 8 per 4096 MFMA cycles per wave).
//   hipcc --offload-arch=gfx950 -O3 experiments/mfma_mix.hip -o experiments/mfma_mix && experiments/mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// SHAPE 0: 32x32x2 (64 cycles), 4 accumulators of 16;  SHAPE 1: 16x16x4 (32 cycles), 16 accumulators of 4
template <int SHAPE, bool ST, bool LD, bool RD, bool SEP = false>
__global__ __launch_bounds__(256) void mix(float* out, const float* src, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[256 * 4 * 9];
  f32x16 a32[4];
  f32x4 a16[16];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) a32[i][r] = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) a16[i][r] = 0.f;
  float a = 1.0f + threadIdx.x, b = 0.5f + threadIdx.x;
  f32x4 v = {a, b, a, b};
  f32x4 fr = {a, b, a, b};
  f32x4 ld = {a, b, a, b};
  const uint32_t waddr = threadIdx.x * 16 + (threadIdx.x >> 3) * 16;
  const float* p = src + (size_t)blockIdx.x * 4096 + threadIdx.x * 4;
  for (int it = 0; it < iters; ++it) {
    // one "k-tile": 4096 MFMA cycles; 8 staging pieces
#pragma unroll
    for (int piece = 0; piece < 8; ++piece) {
      if (SHAPE == 0) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
          asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(a32[u & 3]) : "v"(a), "v"(b));
      } else {
#pragma unroll
        for (int u = 0; u < 16; ++u)
          asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(a16[u]) : "v"(a), "v"(b));
      }
      if (ST) asm volatile("ds_write_b128 %0, %1" : : "v"(waddr), "v"(v) : "memory");
      if (LD && !SEP) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
      if (LD && SEP) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ld) : "v"(p) : "memory");
      if (RD) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(fr) : "v"(waddr) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(fr) : "v"(waddr) : "memory");
      }
    }
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  float s = v.x + fr.x + ld.x;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += a32[i][r];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) s += a16[i][r];
  if (s == 12345.f) out[blockIdx.x * 256 + threadIdx.x] = s + lds[threadIdx.x];
}

template <int SHAPE, bool ST, bool LD, bool RD, bool SEP = false>
static void run(const char* name, float* out, const float* src, int wgs = 512) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 400;
  hipLaunchKernelGGL((mix<SHAPE, ST, LD, RD, SEP>), dim3(wgs), dim3(256), 0, 0, out, src, iters);
  (void)hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0, 0);
    for (int l = 0; l < 20; ++l) hipLaunchKernelGGL((mix<SHAPE, ST, LD, RD, SEP>), dim3(wgs), dim3(256), 0, 0, out, src, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double flops = 20.0 * wgs * 4.0 * iters * 64.0 * 4096.0;
  printf("%-52s wgs %d  %8.3f ms  %7.1f TFLOP/s\n", name, wgs, best, flops / best / 1e9);
}

int main() {
  float *out, *src;
  (void)hipMalloc(&out, 512 * 256 * sizeof(float));
  (void)hipMalloc(&src, 512 * 4096 * sizeof(float));
  (void)hipMemset(src, 0, 512 * 4096 * sizeof(float));
  for (int wgs = 256; wgs <= 512; wgs *= 2) {
    run<0, false, false, false>("32x32x2  bare", out, src, wgs);
    run<0, true, false, false>("32x32x2  + ds_write", out, src, wgs);
    run<0, false, true, false>("32x32x2  + gload", out, src, wgs);
    run<0, false, false, true>("32x32x2  + 2 ds_read", out, src, wgs);
    run<0, true, true, false>("32x32x2  + ds_write + gload (same regs)", out, src, wgs);
    run<0, true, true, false, true>("32x32x2  + ds_write + gload (other regs)", out, src, wgs);
    run<0, true, false, true>("32x32x2  + ds_write + 2 ds_read", out, src, wgs);
    run<0, false, true, true>("32x32x2  + gload + 2 ds_read", out, src, wgs);
    run<0, true, true, true>("32x32x2  + all", out, src, wgs);
    run<0, true, true, true, true>("32x32x2  + all (other regs)", out, src, wgs);
    run<1, false, false, false>("16x16x4  bare", out, src, wgs);
    run<1, true, true, false>("16x16x4  + ds_write + gload (same regs)", out, src, wgs);
    run<1, true, true, true>("16x16x4  + all", out, src, wgs);
    run<1, true, true, true, true>("16x16x4  + all (other regs)", out, src, wgs);
  }
  return 0;
}
