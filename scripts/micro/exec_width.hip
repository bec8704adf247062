// Micro-benchmark: does a wavefront with only 32 / 16 active lanes issue VALU work faster than a full one on gfx950?
// (If it did, single-wave phases such as the PGS sweeps could be split over several lane-sparse waves.)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k(float* out, long long* cyc, int active) {
  const int lane = threadIdx.x;
  float a = lane * 0.001f + 1.f, b = 0.999f, c = 0.5f;
  typedef float f2_t __attribute__((ext_vector_type(2)));
  f2_t p = {a, a + 1.f}, q = {b, b}, r = {c, c};
  unsigned long long t0 = 0, t1 = 0;
  if (lane < active) {
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < 256; it++) {
#pragma unroll
      for (int u = 0; u < 16; u++) { a = a * b + c; p = p * q + r; }   // dependent scalar + packed chains
    }
    t1 = __builtin_amdgcn_s_memtime();
  }
  out[blockIdx.x * 64 + lane] = a + p.x + p.y;
  if (lane == 0) cyc[blockIdx.x] = (long long)(t1 - t0);
}
int main() {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 64 * 64 * 4); (void)hipMalloc(&cyc, 64 * 8);
  long long h[64];
  for (int active : {64, 32, 16, 1}) {
    k<<<64, 64>>>(out, cyc, active);
    (void)hipMemcpy(h, cyc, 64 * 8, hipMemcpyDeviceToHost);
    printf("active lanes %2d: %lld cycles for 4096 fma + 4096 pk_fma (dependent) -> %.2f cycles per instruction\n", active, h[0], h[0] / 8192.0);
  }
  return 0;
}
