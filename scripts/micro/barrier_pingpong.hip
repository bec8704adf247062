// What does one hand-over of a 16-word velocity vector from wave to wave cost?  7 wavefronts of one workgroup (448 threads,
// like k_physics4), round-robin: the owner of step j (wave j % 6) reads 16 words from LDS, does `work` dependent packed
// fmas, writes them back; everybody meets at a workgroup barrier.  Prints cycles per step for: the empty barrier loop, the
// LDS round trip alone, and the full step, with 1 and with 64 workgroups... (one per CU).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/micro/barrier_pingpong scripts/micro/barrier_pingpong.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>   // 0: barrier only, 1: + LDS read/write of 16 words, 2: + 24 dependent-ish packed fmas
__global__ __launch_bounds__(448) void k(long long* out, float* sink, int steps, float seed) {
  __shared__ float sh[32 * 64];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 32 * 64; i += 448) sh[i] = seed * i;
  __syncthreads();
  const int wvs = __builtin_amdgcn_readfirstlane(wv);
  const long long t0 = __builtin_amdgcn_s_memtime();
  f2 acc = {seed, seed};
  for (int j = 0; j < steps; j++) {
    if (wvs == j % 6) {
      if (MODE >= 1) {
        f2 v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = f2{sh[(2 * i) * 64 + lane], sh[(2 * i + 1) * 64 + lane]};
        if (MODE >= 2) {
#pragma unroll
          for (int r = 0; r < 3; r++) {
            f2 a = v[0] * acc;
#pragma unroll
            for (int i = 1; i < 8; i++) a = v[i] * a + acc;
            acc = a * 0.5f;
          }
#pragma unroll
          for (int i = 0; i < 8; i++) v[i] = v[i] * acc.x + acc;
        }
#pragma unroll
        for (int i = 0; i < 8; i++) { sh[(2 * i) * 64 + lane] = v[i].x * 0.999f; sh[(2 * i + 1) * 64 + lane] = v[i].y * 0.999f; }
      }
    }
    __syncthreads();
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (acc.x == 12345.f) sink[0] = acc.y + sh[lane];
}

int main() {
  long long* d; float* s;
  hipMalloc(&d, 256 * sizeof(long long)); hipMalloc(&s, 4);
  const int steps = 2000;
  for (int blocks : {1, 64}) {
    long long h[256];
    k<0><<<blocks, 448>>>(d, s, steps, 0.5f); hipMemcpy(h, d, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    printf("blocks=%d barrier only      : %.1f cycles/step\n", blocks, (double)h[0] / steps);
    k<1><<<blocks, 448>>>(d, s, steps, 0.5f); hipMemcpy(h, d, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    printf("blocks=%d + LDS 16 rd/wr    : %.1f cycles/step\n", blocks, (double)h[0] / steps);
    k<2><<<blocks, 448>>>(d, s, steps, 0.5f); hipMemcpy(h, d, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    printf("blocks=%d + 3x8 pk_fma chain: %.1f cycles/step\n", blocks, (double)h[0] / steps);
  }
  return 0;
}
