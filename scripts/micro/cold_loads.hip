// Micro-benchmark behind the k_post / k_substep memory staging: how long does one workgroup need for a burst of
// ~200 coalesced 256-B row accesses right after a kernel boundary (cold L2), as a function of how the burst is
// spread over wavefronts and of the access width?  Build: hipcc --offload-arch=gfx950 -O3 cold_loads.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define ROWS 200
__global__ void k_write(float* a, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = (float)i;
}
// MODE 0: loads, SoA rows; 1: loads, block-major rows; 2: dwordx4 loads (4 rows per lane, block-major); 3: stores, SoA rows
template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_rw(float* __restrict__ a, int N, int total_rows, long long* out, float* sink) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, b = blockIdx.x, e = b * 64 + lane;
  constexpr int PER = ROWS / WAVES;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  if (MODE == 2) {
    float4 v[PER / 4];
#pragma unroll
    for (int r = 0; r < PER / 4; r++) {
      const int row4 = ((wv * (PER / 4) + r) * 7) % (total_rows / 4);
      v[r] = *(const float4*)(a + (((size_t)b * (total_rows / 4) + row4) * 64 + lane) * 4);
    }
#pragma unroll
    for (int r = 0; r < PER / 4; r++) s += v[r].x + v[r].y + v[r].z + v[r].w;
  } else if (MODE == 3) {
#pragma unroll
    for (int r = 0; r < PER; r++) {
      const int row = ((wv * PER + r) * 7) % total_rows;
      a[(size_t)row * N + e] = (float)r;
    }
    __builtin_amdgcn_s_waitcnt(0);
  } else {
    float v[PER];
#pragma unroll
    for (int r = 0; r < PER; r++) {
      const int row = ((wv * PER + r) * 7) % total_rows;
      v[r] = MODE == 0 ? a[(size_t)row * N + e] : a[((size_t)b * total_rows + row) * 64 + lane];
    }
#pragma unroll
    for (int r = 0; r < PER; r++) s += v[r];
  }
  asm volatile("" ::"v"(s));
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __shared__ long long tmax[WAVES];
  if (lane == 0) tmax[wv] = (long long)(t1 - t0);
  __syncthreads();
  if (threadIdx.x == 0) { long long m = 0; for (int i = 0; i < WAVES; i++) m = m > tmax[i] ? m : tmax[i]; out[b] = m; }
  if (s == 12345.678f) sink[e] = s;
}
int main() {
  const int N = 4096, total_rows = 1400;
  const size_t n = (size_t)N * total_rows;
  float *a, *sink; long long* out;
  (void)hipMalloc(&a, n * 4); (void)hipMalloc(&sink, N * 4); (void)hipMalloc(&out, 64 * 8);
  std::vector<long long> h(64);
  const char* names[] = {"loads SoA 1 wave", "loads SoA 4 waves", "loads block-major 1 wave", "loads block-major 4 waves",
                         "loads x4 block-major 1 wave", "loads x4 block-major 4 waves", "stores SoA 1 wave", "stores SoA 4 waves",
                         "loads SoA 1 wave, producer wrote 1/8", "loads SoA 4 waves, producer wrote 1/8"};
  for (int v = 0; v < 10; v++)
    for (int rep = 0; rep < 3; rep++) {
      k_write<<<1024, 256>>>(a, v >= 8 ? n / 8 : n);      // producer kernel dirties the arena
      switch (v) {
        case 0: case 8: k_rw<0, 1><<<64, 64>>>(a, N, total_rows, out, sink); break;
        case 1: case 9: k_rw<0, 4><<<64, 256>>>(a, N, total_rows, out, sink); break;
        case 2: k_rw<1, 1><<<64, 64>>>(a, N, total_rows, out, sink); break;
        case 3: k_rw<1, 4><<<64, 256>>>(a, N, total_rows, out, sink); break;
        case 4: k_rw<2, 1><<<64, 64>>>(a, N, total_rows, out, sink); break;
        case 5: k_rw<2, 4><<<64, 256>>>(a, N, total_rows, out, sink); break;
        case 6: k_rw<3, 1><<<64, 64>>>(a, N, total_rows, out, sink); break;
        case 7: k_rw<3, 4><<<64, 256>>>(a, N, total_rows, out, sink); break;
      }
      (void)hipMemcpy(h.data(), out, 64 * 8, hipMemcpyDeviceToHost);
      std::sort(h.begin(), h.end());
      if (rep) printf("%-40s rep %d: cycles for %d row accesses per workgroup: min %lld median %lld max %lld\n", names[v], rep, ROWS, h[0], h[32], h[63]);
    }
  return 0;
}
