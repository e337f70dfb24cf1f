// Does the raw-buffer range check of gfx950 include the SGPR offset?  (pipe_mainloop steps k-tiles through soffset and relies on
// rows behind num_records reading as 0.)   hipcc --offload-arch=gfx950 -O3 experiments/buffer_oob.hip -o experiments/buffer_oob
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* base, int num_records, unsigned voff, unsigned soff, float* out) {
  const uint64_t b = reinterpret_cast<uint64_t>(base);
  i32x4 r;
  r.x = (int)(uint32_t)b; r.y = (int)(uint32_t)((b >> 32) & 0xFFFFu); r.z = num_records; r.w = 0x00020000;
  f32x4 v;
  unsigned vo = voff + threadIdx.x * 16;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(vo), "s"(r), "s"(soff) : "memory");
  out[threadIdx.x * 4 + 0] = v.x; out[threadIdx.x * 4 + 1] = v.y; out[threadIdx.x * 4 + 2] = v.z; out[threadIdx.x * 4 + 3] = v.w;
}
int main() {
  float *buf, *out, h[16];
  (void)hipMalloc(&buf, 1 << 20); (void)hipMalloc(&out, 64);
  float* ones = new float[1 << 18];
  for (int i = 0; i < (1 << 18); ++i) ones[i] = 1.0f + i;
  (void)hipMemcpy(buf, ones, 1 << 20, hipMemcpyHostToDevice);
  struct { int nr; unsigned vo, so; const char* what; } cases[] = {
      {4096, 0, 0, "in range"},
      {4096, 4096, 0, "voffset == num_records"},
      {4096, 4080, 0, "last 16 bytes (lanes 1.. out)"},
      {4096, 0, 4096, "soffset == num_records, voffset 0"},
      {4096, 2048, 2048, "voffset + soffset == num_records"},
      {4096, 2048, 2032, "voffset + soffset + 16 == num_records"},
      {4100, 4088, 0, "straddles the end by dwords"},
  };
  for (auto& c : cases) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(4), 0, 0, buf, c.nr, c.vo, c.so, out);
    (void)hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    printf("%-44s nr %5d voff %5u soff %5u -> lane0 %.0f %.0f %.0f %.0f | lane1 %.0f %.0f %.0f %.0f\n", c.what, c.nr, c.vo, c.so, h[0], h[1], h[2], h[3],
           h[4], h[5], h[6], h[7]);
  }
  return 0;
}
