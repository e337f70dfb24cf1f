// Structured buffer (stride = row pitch) with idxen + offen on gfx950: is a load whose byte offset inside the record reaches the
// stride out of range (returns 0)?  That would zero the k tail of a [row][k] operand per ROW for free.
//   hipcc --offload-arch=gfx950 -O3 experiments/buffer_struct_oob.hip -o experiments/buffer_struct_oob
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(const float* base, unsigned stride, unsigned num_records, unsigned index, unsigned voff, unsigned soff, unsigned w3, float* out) {
  const uint64_t b = reinterpret_cast<uint64_t>(base);
  i32x4 r;
  r.x = (int)(uint32_t)b; r.y = (int)((uint32_t)((b >> 32) & 0xFFFFu) | (stride << 16)); r.z = (int)num_records; r.w = (int)w3;
  f32x4 v;
  u32x2 io = {index, voff + threadIdx.x * 16};
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 idxen offen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(io), "s"(r), "s"(soff) : "memory");
  out[threadIdx.x * 4 + 0] = v.x; out[threadIdx.x * 4 + 1] = v.y; out[threadIdx.x * 4 + 2] = v.z; out[threadIdx.x * 4 + 3] = v.w;
}
int main() {
  float *buf, *out, h[16];
  (void)hipMalloc(&buf, 1 << 20); (void)hipMalloc(&out, 64);
  float* v = new float[1 << 18];
  for (int i = 0; i < (1 << 18); ++i) v[i] = (float)i;            // element i of the flat array; row r (pitch 100 floats) starts at 100 r
  (void)hipMemcpy(buf, v, 1 << 20, hipMemcpyHostToDevice);
  struct { unsigned stride, nr, idx, vo, so, w3; const char* what; } cases[] = {
      {400, 1000, 3, 0, 0, 0x00020000, "row 3, offset 0 (expect 300..)"},
      {400, 1000, 3, 384, 0, 0x00020000, "row 3, offset 384: lane0 = last 16 B of the row, lane1 crosses the stride"},
      {400, 1000, 3, 400, 0, 0x00020000, "row 3, offset == stride"},
      {400, 1000, 3, 256, 128, 0x00020000, "row 3, voffset 256 + soffset 128 (= 384): lane1 crosses via soffset?"},
      {400, 1000, 3, 256, 144, 0x00020000, "row 3, voffset 256 + soffset 144 (= 400)"},
      {400, 4, 3, 0, 0, 0x00020000, "last record in range"},
      {400, 4, 4, 0, 0, 0x00020000, "index == num_records"},
  };
  for (auto& c : cases) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(2), 0, 0, buf, c.stride, c.nr, c.idx, c.vo, c.so, c.w3, out);
    (void)hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
    printf("%-78s -> lane0 %.0f %.0f %.0f %.0f | lane1 %.0f %.0f %.0f %.0f\n", c.what, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
  }
  return 0;
}
