// dexsim_device.h -- device-side data layout and small math for the gfx950 kernels.
//
// Data layout in HBM: every per-env quantity is one *field* of the arena, stored SoA as [row][env]
// (env fastest).  A wavefront of 64 lanes = 64 consecutive envs, so every load/store of a field row is one
// fully coalesced 256-byte transaction.  Wave-uniform constants (hand model, config) live in one DevParams
// block that the compiler reads through the scalar cache (s_load), costing no VGPRs.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/dexsim.h"

// (type, name, rows) -- the arena contract.  Rows * num_envs 4-byte words each.
#define DEXSIM_FIELDS(X)                                                                                    \
  /* physics state (replaces PhysX-owned state behind gym.acquire_*_tensor) */                              \
  X(float, q, 26) X(float, qd, 26) X(float, targets, 26)                                                    \
  X(float, box_pos, 3) X(float, box_quat, 4) X(float, box_lin, 3) X(float, box_ang, 3)                      \
  X(float, box_mass, 1) X(float, box_mu, 1)                                                                 \
  /* compact L1 state read by the fused post-physics kernel */                                              \
  X(float, site_pose, 77) X(float, hand_vel, 6) X(float, cforce, 51) X(float, cf5, 15)                      \
  /* dynamics -> contact-solve interface (one sub-step) */                                                  \
  X(float, jframe, 156) X(float, ufree, 32) X(float, fac_sinv, 21) X(float, fac_finv, 60)                   \
  X(float, fac_g, 120) X(int, ncontact, 1) X(float, cgeom, DEXSIM_KMAX * 8) X(int, ccode, DEXSIM_KMAX)      \
  X(float, crow, DEXSIM_KMAX * 3 * 28) X(float, crowq, DEXSIM_KMAX * 3 * 28 + 4) X(float, cbias, DEXSIM_KMAX) X(float, clam, DEXSIM_KMAX * 3) X(float, chdr, DEXSIM_KMAX * 8) X(float, cstage, 6 * NP_STAGE * 9)                                         \
  /* warm-start cache: per contact key one float4 (impulses of the previous sub-step, tag), [key][env][4]; wgen = the env's sub-step generation */ \
  X(float, wlam, DEXSIM_NWKEY * 4) X(int, wgen, 1) X(int, csplit, 1)                                         \
  /* L2 state (ActionProcessor / ObservationEncoder / task / RewardCalculator / TerminationManager) */      \
  X(float, active_prev_targets, 18) X(float, active_rule_targets, 18) X(float, prev_actions, 18)            \
  X(float, actions, 18) X(float, prev_dof_pos, 26)                                                          \
  X(float, contact_duration_steps, 5) X(float, prev_contact_binary, 5) X(int, episode_step, 1)              \
  X(int, success_duration_steps, 1) X(int, success_conditions_met, 1) X(int, current_stage, 1)              \
  X(int, just2, 1) X(int, just3, 1) X(float, time_in_stage, 1) X(float, stage_contact_duration, 1)          \
  X(float, initial_box_pos, 3)                                                                              \
  X(float, prev_finger_dof_vel, 20) X(float, prev_hand_vel, 3) X(float, prev_hand_ang_vel, 3)               \
  X(int, prev_contacts, 5)                                                                                  \
  X(int, episode_success, 1) X(int, episode_failure, 1) X(int, success_reason, DEXSIM_NUM_SUCC)             \
  X(int, failure_reason, DEXSIM_NUM_FAIL) X(int, crit_success, DEXSIM_NUM_SUCC)                             \
  X(int, crit_failure, DEXSIM_NUM_FAIL) X(int, term_success, 1) X(int, term_failure, 1)                     \
  X(int, term_timeout, 1)                                                                                   \
  X(float, obs_all, DEXSIM_OBS_ALL_DIM) X(float, rew_comp, DEXSIM_NUM_REWROWS) X(float, rew, 1)             \
  X(int, reset_flag, 1) X(int, reset_count, 1)

struct Arena {
#define X(type, name, rows) type* name;
  DEXSIM_FIELDS(X)
#undef X
};

// Quad layout (round 3): the per-finger factor hand-off -- fac_g (G_f, 24 words = 6 quads per finger), fac_finv (Fhat_f^-1 lower
// triangle, 10 words in 3 quads per finger) and the finger part of jframe (axes + origins of the finger's 4 joints, 24 words = 6
// quads per finger, behind the 36 row-layout words of the base joints) -- is stored [quad][env][4]: a lane moves four consecutive
// words with ONE 16-byte access.  G_f alone is written and read back by every finger wave in every sub-step: 48 memory instructions
// become 12; the row builder's 58 dependent scalar loads become 15.
#define FQ(field, quad) ((__attribute__((address_space(1))) f4*)(A.field) + ((size_t)(quad) * N + e))
#define JFRAME_FQ0 9   /* first quad of finger 0's joint frames in jframe (36 base-joint words come first) */
// the manifold (cgeom: 2 quads per list entry: p.xyz n.x | n.yz gap mu) and the narrowphase staging blocks (cstage: 2 quads per staged
// contact, 6 waves x 15; their codes as plain rows behind the 180 quads) are FQ fields too
#define NP_STAGE 20                     /* staged entries per narrowphase wave: <= 12 contacts of its three capsules + <= 8 joint-limit rows */
#define CSTAGE_CODE0 (6 * NP_STAGE * 8)

#define CROW_W 28 /* words per contact row: t6 jf4 St6 Fj4 d3 rxd3 Dinv pad */
// crowq: the rows of streamed hand contacts in the quad layout of the LDS row store ([quad][env] of float4, 21 quads per
// contact), followed by ONE always-zero quad row (index KMAX * 21) that lanes without the contact read instead

// per-joint constants packed as one 128-byte record: a wave fetches everything it needs about joint j with two
// s_load_dwordx16 instead of ~12 scattered scalar loads (the base-chain walk was scalar-load-latency bound)
struct JC {
  float qoff[4], poff[3], axis[3], com[3], inertia[6];
  float mass, kp, kd, armature, lo, hi;
  float pad[7];
};
static_assert(sizeof(JC) == 128, "JC must be 128 bytes");

struct DevParams {
  DexSimConfig cfg;
  DexHandModel model;
  float h;           // sub-step
  float box_inv_I_k; // 6 / size^2 : inv inertia of a solid cube = box_inv_I_k / mass
  // closed form of the configuration-independent part of the base chain (joints 0-2 prismatic, joint 3 the first
  // revolute): world axes A_j, the constant part Oc_j of the origin of joint frame j, orientation of frame 3 at q3 = 0
  float base_A[4][3], base_Oc[4][3], base_Q3z[4];
  unsigned obs_div_magic; // ceil(2^32 / num_obs): idx / num_obs == umulhi(idx, magic) for idx < 2^16 (obs_buf flush)
  int inertia_diag;  // every finger link's inertia is diagonal in its joint frame (host-side check of the model): rotate_inertia_diag
  float hand_reach;  // bound on |x - o5| over every point x of every hand capsule, for any joint configuration
                     // (o5 = origin of the palm joint frame): the hand-level broadphase of k_substep
  int obs_col_row[DEXSIM_OBS_ALL_DIM]; // obs_buf column -> obs_all row (flattened policy_observation_keys)
  JC jc[DEXSIM_NJ];  // packed copy of the per-joint model constants
  Arena arena;       // field pointers (filled by dexsim_bind): kernels with long live ranges read them on demand
                     // through the scalar cache instead of pinning 114 SGPRs of by-value kernel arguments
};

// counters block (ints): reduction scratch + device-side control flags
// CNT_ANY_RESET holds a STEP STAMP, not a boolean: the post block of control step `stamp` (a host-side counter, handed to
// every kernel in ApiPtrs::stamp) stores `stamp` when some env must reset; gated launches run iff the word equals their
// stamp.  Nothing ever clears the word, so no launch both clears and sets it (round 1 cleared it in block 0 of the very
// launch whose other blocks set it: an unordered write/atomic pair on a control flag).
#define CNT_ANY_RESET 0
#define CNT_SUCC 1        /* [NUM_SUCC] */
#define CNT_FAIL 2        /* [NUM_FAIL] */
#define CNT_TERM_SUCCESS 8
#define CNT_TERM_FAILURE 9
#define CNT_TERM_TIMEOUT 10
#define CNT_NUM_RESETS 11
#define CNT_CONTACTS 16   /* [2], by stamp parity: sum of active contacts of the control step's main physics step (last
                             sub-step); word stamp&1 is accumulated and read by the step, the other word is cleared for the next */
#define CNT_HAND_CONTACTS 18 /* [2], same scheme: the hand/box and hand/ground contacts among them */
#define CNT_SPIN_TIMEOUT 20  /* diagnostic build -DDEXSIM_DEBUG_SPIN only: (site << 24 | seq) of an LDS token wait that ran into its bound */
#define CNT_PHYS_STEPS 13
#define CNT_CONSECUTIVE 14 /* persistent */
#define CNT_RC_FIRST 15    /* persistent: RewardCalculator lazy prev-state init pending */

// ------------------------------------------------------------------------------------------------- math
typedef float f2 __attribute__((ext_vector_type(2)));   // maps onto the gfx950 packed-FP32 VALU ops
typedef float f4 __attribute__((ext_vector_type(4)));   // 16-byte global accesses
// 16-byte store to global memory (address-space 1: no FLAT instruction even when the pointer came from memory)
__device__ __forceinline__ void st4_global(float* p, f4 v) { *(__attribute__((address_space(1))) f4*)p = v; }
__device__ __forceinline__ void st4_global(float* p, float a, float b, float c, float d) { f4 v; v.x = a; v.y = b; v.z = c; v.w = d; st4_global(p, v); }

struct V3 { float x, y, z; };
struct Q4 { float x, y, z, w; };
struct S6 { float xx, yy, zz, xy, xz, yz; }; // symmetric 3x3
struct M3 { float m[9]; };

#define DI __device__ __forceinline__

DI V3 v3(float x, float y, float z) { return {x, y, z}; }
DI V3 v3p(const float* p) { return {p[0], p[1], p[2]}; }
DI V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
DI V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
DI V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
DI V3& operator+=(V3& a, V3 b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
DI V3& operator-=(V3& a, V3 b) { a.x -= b.x; a.y -= b.y; a.z -= b.z; return a; }
DI float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DI V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
DI float norm(V3 a) { return sqrtf(dot(a, a)); }
DI float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

DI Q4 q4p(const float* p) { return {p[0], p[1], p[2], p[3]}; }
DI Q4 qmul(Q4 a, Q4 b) { // xyzw Hamilton product
  return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
          a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
DI Q4 qconj(Q4 a) { return {-a.x, -a.y, -a.z, a.w}; }
DI M3 q2mat(Q4 q) {
  M3 R;
  float x = q.x, y = q.y, z = q.z, w = q.w;
  R.m[0] = 1 - 2 * (y * y + z * z); R.m[1] = 2 * (x * y - z * w); R.m[2] = 2 * (x * z + y * w);
  R.m[3] = 2 * (x * y + z * w); R.m[4] = 1 - 2 * (x * x + z * z); R.m[5] = 2 * (y * z - x * w);
  R.m[6] = 2 * (x * z - y * w); R.m[7] = 2 * (y * z + x * w); R.m[8] = 1 - 2 * (x * x + y * y);
  return R;
}
DI V3 mul(const M3& R, V3 v) {
  return {R.m[0] * v.x + R.m[1] * v.y + R.m[2] * v.z, R.m[3] * v.x + R.m[4] * v.y + R.m[5] * v.z,
          R.m[6] * v.x + R.m[7] * v.y + R.m[8] * v.z};
}
DI V3 mulT(const M3& R, V3 v) {
  return {R.m[0] * v.x + R.m[3] * v.y + R.m[6] * v.z, R.m[1] * v.x + R.m[4] * v.y + R.m[7] * v.z,
          R.m[2] * v.x + R.m[5] * v.y + R.m[8] * v.z};
}
// quat_rotate_inverse(q, v) = v(2w^2-1) - 2w (q_v x v) + 2 q_v (q_v . v)
DI V3 qrot_inv(Q4 q, V3 v) {
  V3 qv = {q.x, q.y, q.z};
  V3 c = cross(qv, v);
  float d = dot(qv, v), s = 2 * q.w * q.w - 1;
  return {v.x * s - 2 * q.w * c.x + 2 * qv.x * d, v.y * s - 2 * q.w * c.y + 2 * qv.y * d,
          v.z * s - 2 * q.w * c.z + 2 * qv.z * d};
}
DI V3 mul(S6 I, V3 v) {
  return {I.xx * v.x + I.xy * v.y + I.xz * v.z, I.xy * v.x + I.yy * v.y + I.yz * v.z,
          I.xz * v.x + I.yz * v.y + I.zz * v.z};
}
DI S6 operator+(S6 a, S6 b) { return {a.xx + b.xx, a.yy + b.yy, a.zz + b.zz, a.xy + b.xy, a.xz + b.xz, a.yz + b.yz}; }
// R * I_local * R^T for a symmetric local inertia
DI S6 rotate_inertia(const M3& R, const float* s /*xx yy zz xy xz yz*/) {
  float Il[9] = {s[0], s[3], s[4], s[3], s[1], s[5], s[4], s[5], s[2]};
  float T[9];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int k = 0; k < 3; k++) T[3 * r + k] = R.m[3 * r] * Il[k] + R.m[3 * r + 1] * Il[3 + k] + R.m[3 * r + 2] * Il[6 + k];
  auto e = [&](int r, int k) { return T[3 * r] * R.m[3 * k] + T[3 * r + 1] * R.m[3 * k + 1] + T[3 * r + 2] * R.m[3 * k + 2]; };
  return {e(0, 0), e(1, 1), e(2, 2), e(0, 1), e(0, 2), e(1, 2)};
}
// R * diag(a, b, c) * R^T: link inertias given in their principal frame (DevParams::inertia_diag: every link of the model) -- 27
// instead of 45 multiply-adds, four times per finger wave and sub-step
DI S6 rotate_inertia_diag(const M3& R, const float* s /*xx yy zz*/) {
  const float a0 = s[0] * R.m[0], a3 = s[0] * R.m[3], a6 = s[0] * R.m[6];
  const float b1 = s[1] * R.m[1], b4 = s[1] * R.m[4], b7 = s[1] * R.m[7];
  const float c2 = s[2] * R.m[2], c5 = s[2] * R.m[5], c8 = s[2] * R.m[8];
  return {a0 * R.m[0] + b1 * R.m[1] + c2 * R.m[2], a3 * R.m[3] + b4 * R.m[4] + c5 * R.m[5], a6 * R.m[6] + b7 * R.m[7] + c8 * R.m[8],
          a0 * R.m[3] + b1 * R.m[4] + c2 * R.m[5], a0 * R.m[6] + b1 * R.m[7] + c2 * R.m[8], a3 * R.m[6] + b4 * R.m[7] + c5 * R.m[8]};
}
// parallel-axis term  m (|d|^2 E - d d^T)
DI S6 pa(float m, V3 d) {
  float dd = dot(d, d);
  return {m * (dd - d.x * d.x), m * (dd - d.y * d.y), m * (dd - d.z * d.z), -m * d.x * d.y, -m * d.x * d.z, -m * d.y * d.z};
}

struct Comp { float m; V3 c; S6 I; };
// a <- a (+) b  (both with positive or zero mass; zero+zero stays at a)
DI void comp_add(Comp& a, const Comp& b) {
  float m = a.m + b.m;
  if (m <= 0.f) return;
  float inv = 1.f / m;
  V3 c = {(a.m * a.c.x + b.m * b.c.x) * inv, (a.m * a.c.y + b.m * b.c.y) * inv, (a.m * a.c.z + b.m * b.c.z) * inv};
  // the two parallel-axis shifts to the common centre of mass are one shift with the reduced mass: m_a |c_a - c|^2 (x) +
  // m_b |c_b - c|^2 (x) = (m_a m_b / m) |c_a - c_b|^2 (x) -- half the instructions, three times per finger and sub-step
  S6 I = a.I + b.I + pa(a.m * b.m * inv, a.c - b.c);
  a.m = m; a.c = c; a.I = I;
}

// in-place inverse of an SPD n x n matrix (row-major full storage) through Cholesky; fully unrolled
template <int n>
DI void spd_inverse(float* A) {
  float L[n * n];
#pragma unroll
  for (int i = 0; i < n; i++)
#pragma unroll
    for (int j = 0; j <= i; j++) {
      float s = A[i * n + j];
#pragma unroll
      for (int k = 0; k < j; k++) s -= L[i * n + k] * L[j * n + k];
      if (i == j) L[i * n + i] = sqrtf(fmaxf(s, 1e-20f));
      else L[i * n + j] = s / L[j * n + j];
    }
  float Li[n * n]; // inverse of L (lower)
#pragma unroll
  for (int c = 0; c < n; c++)
#pragma unroll
    for (int i = c; i < n; i++) {
      float s = (i == c) ? 1.f : 0.f;
#pragma unroll
      for (int k = c; k < i; k++) s -= L[i * n + k] * Li[k * n + c];
      Li[i * n + c] = s / L[i * n + i];
    }
#pragma unroll
  for (int i = 0; i < n; i++)
#pragma unroll
    for (int j = 0; j <= i; j++) {
      float s = 0.f;
#pragma unroll
      for (int k = i; k < n; k++) s += Li[k * n + i] * Li[k * n + j];
      A[i * n + j] = s;
      A[j * n + i] = s;
    }
}

// x = A^-1 b for an SPD n x n matrix (row-major, lower triangle read) through Cholesky and two triangular solves; fully
// unrolled.  About 40 % of the arithmetic of spd_inverse + matrix-vector product: the Schur phase of a sub-step without hand
// contacts needs u_B only, not S^-1.
template <int n>
DI void spd_solve(const float* A, const float* b, float* x) {
  float L[n * n], rd[n];   // rd = 1 / L_ii
#pragma unroll
  for (int i = 0; i < n; i++)
#pragma unroll
    for (int j = 0; j <= i; j++) {
      float s = A[i * n + j];
#pragma unroll
      for (int k = 0; k < j; k++) s -= L[i * n + k] * L[j * n + k];
      if (i == j) { const float d = sqrtf(fmaxf(s, 1e-20f)); L[i * n + i] = d; rd[i] = 1.f / d; }
      else L[i * n + j] = s * rd[j];
    }
  float y[n];
#pragma unroll
  for (int i = 0; i < n; i++) {
    float s = b[i];
#pragma unroll
    for (int k = 0; k < i; k++) s -= L[i * n + k] * y[k];
    y[i] = s * rd[i];
  }
#pragma unroll
  for (int i = n - 1; i >= 0; i--) {
    float s = y[i];
#pragma unroll
    for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k];
    x[i] = s * rd[i];
  }
}

DI void tangent_basis(V3 n, V3& t1, V3& t2) {
  V3 e = (fabsf(n.x) < 0.57735f) ? v3(1, 0, 0) : v3(0, 1, 0);
  t1 = cross(e, n);
  float l = norm(t1);
  t1 = {t1.x / l, t1.y / l, t1.z / l};
  t2 = cross(n, t1);
}

// sin and cos of a joint half-angle.  libm's sincosf costs ~145 instructions on gfx950 (its Payne-Hanek large-argument
// reduction is inlined branch-free) and sits 7 times on every wave's critical path per sub-step; joint angles are
// bounded, so one Cody-Waite step to [-pi/4, pi/4] and the classic single-precision minimax kernels (Cephes sinf /
// cosf coefficients, < 1 ulp on the reduced range) do the same job in ~25.  Valid to full accuracy for |h| < ~100.
__device__ __forceinline__ void sincos_joint(float h, float* sn, float* cs) {
  const float kf = rintf(h * 0.6366197723675814f);            // nearest multiple of pi/2
  float r = fmaf(kf, -1.5707963705062866f, h);                 // pi/2 split in two floats (Cody-Waite)
  r = fmaf(kf, 4.371139000186241e-8f, r);
  const float z = r * r;
  const float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * r, r);
  const float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z,
                        fmaf(-0.5f, z, 1.0f));
  const int k = (int)kf;
  const bool swap = (k & 1) != 0;
  const float s0 = swap ? pc : ps, c0 = swap ? ps : pc;
  *sn = (k & 2) ? -s0 : s0;
  *cs = ((k + 1) & 2) ? -c0 : c0;
}

// Philox4x32-10 (device-side reset / domain-randomisation stream)
DI void philox4x32(uint32_t out[4], uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  uint32_t c[4] = {c0, c1, c2, c3};
#pragma unroll
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) out[i] = c[i];
}
DI float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }
