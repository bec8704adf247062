// Would a 4-lanes-per-env layout shorten the per-env dependent chain?  
//
// Today one lane = one env: a 3-vector operation is 3 instructions on a wave that serves 64 envs, and at N = 4096 only
// 64 workgroups exist (64 of 256 CUs busy).  The alternative: one QUAD of lanes per env (x, y, z, w of every vector /
// quaternion in lanes 0..3 of the quad, cross products and quaternion products through DPP quad_perm), 16 envs per wave,
// 4x the workgroups.  What matters for the control-step time is then the length of the dependent chain per env.
//
// This micro-benchmark runs the core of finger_dynamics' first loop -- the FK / RNEA-forward chain of a 4-joint finger:
// per joint  on = o + R(q) poff;  qz = q (x) qoff;  aw = R(qz) axis;  qn = qz (x) (axis sin, cos);  r = on - o;
//            ao += al x r + w x (w x r);  al += qd (w x aw);  w += qd aw;  c = on + R(qn) com
// in both layouts on identical data, checks that they agree, and prints cycles per chain (s_memtime) with one wave per
// SIMD, plus the instruction counts of the two loops (from the disassembly: llvm-objdump -d on the executable).
//
//   hipcc --offload-arch=gfx950 -O3 -o scripts/micro/quad_layout scripts/micro/quad_layout.hip && ./scripts/micro/quad_layout
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

#define NJ 4
struct JointC { float qoff[4], poff[3], axis[3], com[3]; };
__constant__ JointC c_j[NJ];

// ------------------------------------------------------------------------------------------------ layout A: lane = env
struct V3 { float x, y, z; };
struct Q4 { float x, y, z, w; };
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ Q4 qmul(Q4 a, Q4 b) {
  return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
          a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
__device__ __forceinline__ V3 qrot(Q4 q, V3 v) {   // v + 2 w (qv x v) + 2 qv x (qv x v)
  const V3 qv = {q.x, q.y, q.z};
  const V3 t = 2.f * cross(qv, v);
  return v + q.w * t + cross(qv, t);
}
__device__ __forceinline__ void sincos_small(float h, float* s, float* c) {   // |h| < 1: enough for the benchmark
  const float z = h * h;
  *s = h * (1.f + z * (-1.f / 6 + z * (1.f / 120)));
  *c = 1.f + z * (-0.5f + z * (1.f / 24 - z * (1.f / 720)));
}

__global__ __launch_bounds__(64) void k_scalar(const float* __restrict__ qin, float* __restrict__ out, long long* cyc, int reps) {
  const int e = blockIdx.x * 64 + threadIdx.x;
  float ql[NJ], qdl[NJ];
  for (int l = 0; l < NJ; l++) { ql[l] = qin[l * gridDim.x * 64 + e]; qdl[l] = 0.3f * ql[l]; }
  V3 acc = {0, 0, 0};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < reps; rep++) {
    Q4 qf = {0, 0, 0, 1}; V3 of = {acc.x * 1e-9f, 0, 0.5f}, wf = {0, 0, 0}, alf = {0, 0, 0}, aof = {0, 0, 9.81f};
    V3 csum = {0, 0, 0};
    const float fb = 1e-7f * acc.x;   // every repetition depends on the previous one: nothing can be hoisted out of the loop
#pragma unroll
    for (int l = 0; l < NJ; l++) {
      const JointC& J = c_j[l];
      const V3 on = of + qrot(qf, V3{J.poff[0], J.poff[1], J.poff[2]});
      const Q4 qz = qmul(qf, Q4{J.qoff[0], J.qoff[1], J.qoff[2], J.qoff[3]});
      const V3 ax = {J.axis[0], J.axis[1], J.axis[2]};
      const V3 aw = qrot(qz, ax);
      float s, c;
      sincos_small(0.5f * (ql[l] + fb), &s, &c);
      const Q4 qn = qmul(qz, Q4{ax.x * s, ax.y * s, ax.z * s, c});
      const V3 r = on - of;
      aof = aof + cross(alf, r) + cross(wf, cross(wf, r));
      alf = alf + (qdl[l] + fb) * cross(wf, aw);
      wf = wf + (qdl[l] + fb) * aw;
      csum = csum + on + qrot(qn, V3{J.com[0], J.com[1], J.com[2]});
      qf = qn; of = on;
    }
    acc = acc + csum + aof + alf + wf;
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[e * 3] = acc.x; out[e * 3 + 1] = acc.y; out[e * 3 + 2] = acc.z;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// ------------------------------------------------------------------------------------------------ layout B: quad = env
// lane c of a quad holds component c (x, y, z, w); 3-vectors keep w = 0
template <int CTRL> __device__ __forceinline__ float qp(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
#define QP_YZX 0xC9   /* [1,2,0,3] */
#define QP_ZXY 0xD2   /* [2,0,1,3] */
#define QP_WWW 0xFF   /* [3,3,3,3] */
#define QP_SW1 0xB1   /* [1,0,3,2] */
#define QP_SW2 0x4E   /* [2,3,0,1] */
__device__ __forceinline__ float qcross(float a, float b) { return qp<QP_YZX>(a) * qp<QP_ZXY>(b) - qp<QP_ZXY>(a) * qp<QP_YZX>(b); }
__device__ __forceinline__ float qsum(float p) { p += qp<QP_SW1>(p); return p + qp<QP_SW2>(p); }   // all 4 lanes, result in every lane
// quaternion product a (x) b, xyzw; m3 = 1 in lanes 0..2, 0 in lane 3
__device__ __forceinline__ float qqmul(float a, float b, float m3) {
  const float aw = qp<QP_WWW>(a), bw = qp<QP_WWW>(b);
  const float av = a * m3, bv = b * m3;                       // vector parts (w lane zero)
  const float d = qsum(av * bv);                              // a.xyz . b.xyz
  return aw * b + m3 * (bw * a + qcross(av, bv)) - (1.f - m3) * d;
}
// rotate the 3-vector v (w lane 0) by the quaternion q
__device__ __forceinline__ float qqrot(float q, float v, float m3) {
  const float qv = q * m3, qw = qp<QP_WWW>(q);
  const float t = 2.f * qcross(qv, v);
  return v + qw * t + qcross(qv, t);
}

__global__ __launch_bounds__(64) void k_quad(const float* __restrict__ qin, float* __restrict__ out, long long* cyc, int reps, int nenv) {
  const int lane = threadIdx.x, c = lane & 3;
  const int e = blockIdx.x * 16 + (lane >> 2);
  const float m3 = c < 3 ? 1.f : 0.f;
  float ql[NJ], qdl[NJ];
  for (int l = 0; l < NJ; l++) { ql[l] = qin[l * nenv + e]; qdl[l] = 0.3f * ql[l]; }
  // per-lane constants of the joints: component c of qoff / poff / axis / com
  float kq[NJ], kp[NJ], ka[NJ], kc[NJ];
  for (int l = 0; l < NJ; l++) {
    kq[l] = c_j[l].qoff[c];
    kp[l] = c < 3 ? c_j[l].poff[c] : 0.f; ka[l] = c < 3 ? c_j[l].axis[c] : 0.f; kc[l] = c < 3 ? c_j[l].com[c] : 0.f;
  }
  float acc = 0.f;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < reps; rep++) {
    const float accx = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(acc), 0x00, 0xf, 0xf, true));   // lane 0 of the quad
    float qf = c == 3 ? 1.f : 0.f, of = c == 0 ? accx * 1e-9f : (c == 2 ? 0.5f : 0.f), wf = 0.f, alf = 0.f, aof = c == 2 ? 9.81f : 0.f;
    float csum = 0.f;
    const float fb = 1e-7f * accx;
#pragma unroll
    for (int l = 0; l < NJ; l++) {
      const float on = of + qqrot(qf, kp[l], m3);
      const float qz = qqmul(qf, kq[l], m3);
      const float aw = qqrot(qz, ka[l], m3);
      float s, co;
      sincos_small(0.5f * (ql[l] + fb), &s, &co);
      const float qj = c < 3 ? ka[l] * s : co;
      const float qn = qqmul(qz, qj, m3);
      const float r = on - of;
      aof = aof + qcross(alf, r) + qcross(wf, qcross(wf, r));
      alf = alf + (qdl[l] + fb) * qcross(wf, aw);
      wf = wf + (qdl[l] + fb) * aw;
      csum = csum + on + qqrot(qn, kc[l], m3);
      qf = qn; of = on;
    }
    acc = acc + csum + aof + alf + wf;
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (c < 3) out[e * 3 + c] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  const int nenv = 4096, reps = 200;
  JointC hj[NJ];
  for (int l = 0; l < NJ; l++) {
    const float a = 0.3f + 0.1f * l, sn = std::sin(a / 2), cs = std::cos(a / 2);
    hj[l] = JointC{{sn * 0.6f, sn * 0.0f, sn * 0.8f, cs}, {0.01f * l, 0.03f, 0.02f}, {0.f, 1.f, 0.f}, {0.0f, 0.015f, 0.002f * l}};
    if (l & 1) { hj[l].axis[0] = 1.f; hj[l].axis[1] = 0.f; }
  }
  CK(hipMemcpyToSymbol(HIP_SYMBOL(c_j), hj, sizeof hj));
  std::vector<float> hq(NJ * nenv);
  for (size_t i = 0; i < hq.size(); i++) hq[i] = 0.9f * std::sin(0.37f * (float)i);
  float *dq, *oa, *ob; long long* dc;
  CK(hipMalloc(&dq, hq.size() * 4)); CK(hipMalloc(&oa, nenv * 12)); CK(hipMalloc(&ob, nenv * 12)); CK(hipMalloc(&dc, 256 * 8));
  CK(hipMemcpy(dq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
  long long ha[256], hb[256];
  k_scalar<<<nenv / 64, 64>>>(dq, oa, dc, reps);
  CK(hipMemcpy(ha, dc, (nenv / 64) * 8, hipMemcpyDeviceToHost));
  k_quad<<<nenv / 16, 64>>>(dq, ob, dc, reps, nenv);
  CK(hipMemcpy(hb, dc, 256 * 8, hipMemcpyDeviceToHost));
  std::vector<float> va(nenv * 3), vb(nenv * 3);
  CK(hipMemcpy(va.data(), oa, nenv * 12, hipMemcpyDeviceToHost));
  CK(hipMemcpy(vb.data(), ob, nenv * 12, hipMemcpyDeviceToHost));
  double err = 0, mag = 0;
  for (int i = 0; i < nenv * 3; i++) { err = std::fmax(err, std::fabs((double)va[i] - vb[i])); mag = std::fmax(mag, std::fabs((double)va[i])); }
  printf("agreement: max |scalar - quad| = %.3e (values up to %.3e)\n", err, mag);
  printf("lane = env  (64 envs/wave,  %3d workgroups): %.1f cycles per 4-joint chain\n", nenv / 64, (double)ha[0] / reps);
  printf("quad = env  (16 envs/wave,  %3d workgroups): %.1f cycles per 4-joint chain\n", nenv / 16, (double)hb[0] / reps);
  printf("chain speed-up of the quad layout: %.2fx\n", (double)ha[0] / (double)hb[0]);
  return 0;
}
