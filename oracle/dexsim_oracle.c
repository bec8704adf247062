/*
 * dexsim_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.  The product
 * (dexrobot_isaac_amd + libdexsim) never links, imports or executes anything in oracle/.
 *
 * What it restates, in plain scalar C, one env at a time:
 *   L2 (post-physics tensor math) -- a line-by-line restatement of the reference's Python:
 *     ActionProcessor.process_actions      dexhand_env/components/action/action_processor.py:284-352
 *       position / position_delta rules    components/action/default_rules.py:33-112, scaling.py:28-99
 *       velocity_clamp, position_clamp     components/action/rules.py:141-190
 *       apply_coupling                     action_processor.py:571-614
 *       extract_active_targets_from_full   action_processor.py:616-666
 *     ObservationEncoder                   components/observation/observation_encoder.py:298-361,576-758,915-986,1483-1521
 *     BlindGraspingTask                    tasks/blind_grasping_task.py:433-547 (reset), 549-978 (obs+FSM),
 *                                          980-1208 (rewards), 1210-1364 (success/failure)
 *     StepProcessor                        components/step_processor.py:37-232
 *     TerminationManager                   components/termination/termination_manager.py:120-357
 *     RewardCalculator                     components/reward/reward_calculator.py:66-275
 *     ResetManager / DexHandBase.reset*    components/reset/reset_manager.py:92-190, tasks/dexhand_base.py:743-838
 *   This part is PINNED by golden vectors generated from the reference's own Python (tests/golden/).
 *
 *   L0/L1 (physics) -- the reference delegates to Isaac Gym Preview 4 / PhysX (closed binary, absent;
 *     call sites components/physics/physics_manager.py:73-119).  There is no arithmetic to restate, so this
 *     file is the *specification* of the build's own integrator in its most transparent form: generic-tree
 *     CRBA into a dense 26x26 mass matrix, dense Cholesky, dense M^-1 J^T, PGS in plain generalized-velocity
 *     space.  The HIP kernels implement the same model with a block-structured (Schur complement) solver;
 *     agreement between the two is the parity test.  PARITY vs PhysX: UNPINNED (no reference test or binary
 *     pins it).
 *
 * Arithmetic is fp32 (`real` = float) so the oracle is an fp32 restatement like the reference's torch
 * path; compile with -DORC_DOUBLE for an fp64 variant used to size the fp32 tolerance.
 */
#include "../include/dexsim.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef ORC_DOUBLE
typedef double real;
#define RSQRT sqrt
#define RSIN sin
#define RCOS cos
#define REXP exp
#define RFABS fabs
#else
typedef float real;
#define RSQRT sqrtf
#define RSIN sinf
#define RCOS cosf
#define REXP expf
#define RFABS fabsf
#endif

#define NJ DEXSIM_NJ
#define NV 32 /* 26 hand DOFs + 6 box twist */
#define KMAX DEXSIM_KMAX

/* ------------------------------------------------------------------------------------------ small math */
static inline real clampr(real x, real lo, real hi) { return x < lo ? lo : (x > hi ? hi : x); }
static inline real minr(real a, real b) { return a < b ? a : b; }
static inline real maxr(real a, real b) { return a > b ? a : b; }
static inline void v3set(real* o, real x, real y, real z) { o[0] = x; o[1] = y; o[2] = z; }
static inline void v3cpy(real* o, const real* a) { o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; }
static inline void v3add(real* o, const real* a, const real* b) { o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2]; }
static inline void v3sub(real* o, const real* a, const real* b) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
static inline void v3axpy(real* o, real s, const real* a) { o[0] += s * a[0]; o[1] += s * a[1]; o[2] += s * a[2]; }
static inline real v3dot(const real* a, const real* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void v3cross(real* o, const real* a, const real* b) {
  real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
static inline real v3norm(const real* a) { return RSQRT(v3dot(a, a)); }

/* xyzw Hamilton product (isaacgym.torch_utils.quat_mul semantics; source absent, pinned by the
 * known-answer cases of reference utils/test_coordinate_transforms.py through tests/golden) */
static inline void qmul(real* o, const real* a, const real* b) {
  real x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  real y = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  real z = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
  real w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}
static inline void qconj(real* o, const real* a) { o[0] = -a[0]; o[1] = -a[1]; o[2] = -a[2]; o[3] = a[3]; }
/* quat_rotate_inverse(q, v) = v(2w^2-1) - 2w (q_v x v) + 2 q_v (q_v . v)    */
static inline void qrot_inv(real* o, const real* q, const real* v) {
  real w = q[3], c[3], d = v3dot(q, v);
  v3cross(c, q, v);
  real s = 2 * w * w - 1;
  for (int i = 0; i < 3; i++) o[i] = v[i] * s - 2 * w * c[i] + 2 * q[i] * d;
}
static inline void qrot(real* o, const real* q, const real* v) {
  real w = q[3], c[3], d = v3dot(q, v);
  v3cross(c, q, v);
  real s = 2 * w * w - 1;
  for (int i = 0; i < 3; i++) o[i] = v[i] * s + 2 * w * c[i] + 2 * q[i] * d;
}
static inline void q2mat(real* R, const real* q) {
  real x = q[0], y = q[1], z = q[2], w = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}
static inline void m3v(real* o, const real* R, const real* v) {
  real x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
  real y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
  real z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static inline void m3tv(real* o, const real* R, const real* v) {
  real x = R[0] * v[0] + R[3] * v[1] + R[6] * v[2];
  real y = R[1] * v[0] + R[4] * v[1] + R[7] * v[2];
  real z = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}

/* ------------------------------------------------------------------------------------------ Philox4x32-10 */
static inline void philox_round(uint32_t* c, uint32_t k0, uint32_t k1) {
  uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
  uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
static void philox4x32(uint32_t out[4], uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  uint32_t c[4] = {c0, c1, c2, c3};
  for (int r = 0; r < 10; r++) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  for (int i = 0; i < 4; i++) out[i] = c[i];
}
/* torch.rand semantics: 24 random mantissa bits -> [0,1) */
static inline real u01(uint32_t x) { return (real)(x >> 8) * (real)(1.0 / 16777216.0); }

/* ------------------------------------------------------------------------------------------ state */
typedef struct OrcContact {
  real p[3], n[3], gap, mu;
  int type; /* 0 hand-ground, 1 hand-box, 2 box-ground, 3 joint-limit row (cfg.joint_limit_rows) */
  int cap;  /* capsule index for hand contacts; type 3: the joint's level in its finger (0..3) */
  int key;  /* warm-start cache: hand contacts slot (cap * 2 + type) * 2 + sample; box/ground: the corner index (slot 80 + list index);
             * type 3: slot 88 + (finger joint * 2 + side) */
  int finger, side;   /* type 3: finger 0..4, side 0 = lower limit (J = +e_j), 1 = upper limit (J = -e_j) */
} OrcContact;

typedef struct OrcEnv {
  /* physics state */
  real q[NJ], qd[NJ], targets[NJ];
  real box_pos[3], box_quat[4], box_lin[3], box_ang[3], box_mass, box_mu;
  /* L1 compact state published after each physics step */
  real site_pose[DEXSIM_NSITE][7];
  real hand_vel[6];
  real cforce[DEXSIM_NFSLOT][3];
  /* TensorManager.contact_forces: the (N,5,3) gather COPY made by refresh_tensors at the start of
   * process_physics_step (tensor_manager.py:441-446, step_processor.py:46-48).  The first observation pass
   * of DexHandBase.reset() runs before that refresh and therefore sees the previous copy. */
  real cf5[5][3];
  int ncontact;
  OrcContact contact[KMAX];
  /* warm-start cache: the impulses a contact ended the previous sub-step with.  Hand contacts: slot = (capsule, type, sample),
   * valid while tag == 8 * wgen; box/ground contacts: slot 80 + list index k (k < 4), valid while tag == 8 * wgen + corner.
   * wgen = the env's sub-step generation (advanced by every sub-step and by a reset) */
  real wlam[DEXSIM_NWKEY][3];
  int wtag[DEXSIM_NWKEY], wgen;
  /* L2 state */
  real active_prev_targets[18], active_rule_targets[18], prev_actions[18], actions[18], prev_dof_pos[NJ];
  real contact_duration_steps[5], prev_contact_binary[5];
  int episode_step;
  int success_duration_steps, success_conditions_met, current_stage, just2, just3;
  real time_in_stage, stage_contact_duration, initial_box_pos[3];
  real prev_finger_dof_vel[20], prev_hand_vel[3], prev_hand_ang_vel[3];
  int prev_contacts[5];
  int episode_success, episode_failure, success_reason[DEXSIM_NUM_SUCC], failure_reason[DEXSIM_NUM_FAIL];
  int crit_success[DEXSIM_NUM_SUCC], crit_failure[DEXSIM_NUM_FAIL], term_success, term_failure, term_timeout;
  real obs_all[DEXSIM_OBS_ALL_DIM];
  real rew_comp[DEXSIM_NUM_REWROWS];
  real rew;
  int reset_flag, reset_count;
} OrcEnv;

typedef struct Oracle {
  DexSimConfig cfg;
  DexHandModel model;
  int n;
  OrcEnv* env;
  float* obs_buf;          /* (N, num_obs) */
  float* reset_samples;    /* (N, 29) or NULL */
  float stats[DEXSIM_STAT_WORDS];
  int rc_first_call;       /* RewardCalculator prev-state lazy init (reward_calculator.py:146-150,189-191) */
  int consecutive_successes;
  int any_reset;
  int nthreads;
} Oracle;

/* ------------------------------------------------------------------------------------------ kinematics */
typedef struct FK {
  real q[NJ][4]; /* joint frame orientation (after joint motion), xyzw */
  real R[NJ][9];
  real o[NJ][3]; /* joint frame origin */
  real a[NJ][3]; /* joint axis, world */
} FK;

static inline int joint_parent(int j) { return j == 0 ? -1 : (j < 6 ? j - 1 : (((j - 6) & 3) == 0 ? 5 : j - 1)); }

static void fk_compute(const DexHandModel* m, const real* q, FK* fk) {
  for (int j = 0; j < NJ; j++) {
    int p = joint_parent(j);
    real qp[4], op[3], Rp[9];
    if (p < 0) {
      for (int i = 0; i < 4; i++) qp[i] = m->spawn_quat[i];
      for (int i = 0; i < 3; i++) op[i] = m->spawn_pos[i];
    } else {
      memcpy(qp, fk->q[p], sizeof qp);
      memcpy(op, fk->o[p], sizeof op);
    }
    q2mat(Rp, qp);
    real poff[3] = {m->jpoff[j][0], m->jpoff[j][1], m->jpoff[j][2]}, t[3];
    m3v(t, Rp, poff);
    v3add(fk->o[j], op, t);
    real qoff[4] = {m->jqoff[j][0], m->jqoff[j][1], m->jqoff[j][2], m->jqoff[j][3]}, qz[4], Rz[9];
    qmul(qz, qp, qoff);
    q2mat(Rz, qz);
    real ax[3] = {m->jaxis[j][0], m->jaxis[j][1], m->jaxis[j][2]};
    m3v(fk->a[j], Rz, ax);
    if (m->jtype[j] == 0) {
      memcpy(fk->q[j], qz, sizeof qz);
      v3axpy(fk->o[j], q[j], fk->a[j]);
    } else {
      real h = (real)0.5 * q[j], s = RSIN(h), qj[4] = {ax[0] * s, ax[1] * s, ax[2] * s, RCOS(h)};
      qmul(fk->q[j], qz, qj);
    }
    q2mat(fk->R[j], fk->q[j]);
  }
}

/* body j: world COM and world inertia (about COM) */
static void body_world(const DexHandModel* m, const FK* fk, int j, real* c, real* I /*3x3*/) {
  real com[3] = {m->com[j][0], m->com[j][1], m->com[j][2]}, t[3];
  m3v(t, fk->R[j], com);
  v3add(c, fk->o[j], t);
  const float* s = m->inertia[j];
  real Il[9] = {s[0], s[3], s[4], s[3], s[1], s[5], s[4], s[5], s[2]};
  const real* R = fk->R[j];
  real T[9];
  for (int r = 0; r < 3; r++)
    for (int k = 0; k < 3; k++) T[3 * r + k] = R[3 * r] * Il[k] + R[3 * r + 1] * Il[3 + k] + R[3 * r + 2] * Il[6 + k];
  for (int r = 0; r < 3; r++)
    for (int k = 0; k < 3; k++) I[3 * r + k] = T[3 * r] * R[3 * k] + T[3 * r + 1] * R[3 * k + 1] + T[3 * r + 2] * R[3 * k + 2];
}

/* point Jacobian column of joint j for world point p */
static inline void jac_col(const DexHandModel* m, const FK* fk, int j, const real* p, real* jv) {
  if (m->jtype[j] == 0) v3cpy(jv, fk->a[j]);
  else { real r[3]; v3sub(r, p, fk->o[j]); v3cross(jv, fk->a[j], r); }
}

/* ------------------------------------------------------------------------------------------ dynamics (dense) */
typedef struct Comp { real m, c[3], I[9]; } Comp;

static void comp_add(Comp* a, const Comp* b) { /* a <- a (+) b : parallel-axis combination about the joint COM */
  real m = a->m + b->m;
  if (m <= 0) return;
  real c[3];
  for (int i = 0; i < 3; i++) c[i] = (a->m * a->c[i] + b->m * b->c[i]) / m;
  real I[9];
  for (int i = 0; i < 9; i++) I[i] = a->I[i] + b->I[i];
  const Comp* s[2] = {a, b};
  for (int k = 0; k < 2; k++) {
    real d[3]; v3sub(d, s[k]->c, c);
    real dd = v3dot(d, d), mk = s[k]->m;
    for (int r = 0; r < 3; r++)
      for (int cc = 0; cc < 3; cc++) I[3 * r + cc] += mk * ((r == cc ? dd : 0) - d[r] * d[cc]);
  }
  a->m = m; v3cpy(a->c, c); memcpy(a->I, I, sizeof I);
}

/* M (26x26, symmetric) by the composite-rigid-body method in momentum form */
static void crba(const DexHandModel* m, const FK* fk, real M[NJ][NJ]) {
  Comp comp[NJ];
  for (int j = 0; j < NJ; j++) {
    comp[j].m = m->mass[j];
    body_world(m, fk, j, comp[j].c, comp[j].I);
    if (comp[j].m <= 0) { v3cpy(comp[j].c, fk->o[j]); memset(comp[j].I, 0, sizeof comp[j].I); }
  }
  for (int j = NJ - 1; j > 0; j--) {
    int p = joint_parent(j);
    if (comp[p].m <= 0 && comp[j].m > 0) { Comp t = comp[j]; comp[p] = t; }
    else comp_add(&comp[p], &comp[j]);
  }
  memset(M, 0, sizeof(real) * NJ * NJ);
  for (int j = 0; j < NJ; j++) {
    /* momentum of subtree j under unit velocity of joint j */
    real P[3], L[3];
    if (m->jtype[j] == 0) {
      for (int i = 0; i < 3; i++) { P[i] = comp[j].m * fk->a[j][i]; L[i] = 0; }
    } else {
      real r[3], v[3];
      v3sub(r, comp[j].c, fk->o[j]);
      v3cross(v, fk->a[j], r);
      for (int i = 0; i < 3; i++) P[i] = comp[j].m * v[i];
      m3v(L, comp[j].I, fk->a[j]);
    }
    for (int k = j; k >= 0; k = joint_parent(k)) {
      real val;
      if (m->jtype[k] == 0) val = v3dot(fk->a[k], P);
      else {
        real r[3], rxP[3], t[3];
        v3sub(r, comp[j].c, fk->o[k]);
        v3cross(rxP, r, P);
        v3add(t, L, rxP);
        val = v3dot(fk->a[k], t);
      }
      M[j][k] = M[k][j] = val;
    }
  }
}

/* bias forces (gravity + velocity-product terms) by recursive Newton-Euler with qdd = 0 */
static void rnea_bias(const DexHandModel* m, const FK* fk, const real* qd, const real* g, real* tau) {
  real w[NJ][3], al[NJ][3], ao[NJ][3], vo_rel_unused;
  real f[NJ][3], n[NJ][3];
  (void)vo_rel_unused;
  for (int j = 0; j < NJ; j++) {
    int p = joint_parent(j);
    real wp[3] = {0, 0, 0}, alp[3] = {0, 0, 0}, aop[3] = {-g[0], -g[1], -g[2]}, op[3];
    if (p >= 0) { v3cpy(wp, w[p]); v3cpy(alp, al[p]); v3cpy(aop, ao[p]); v3cpy(op, fk->o[p]); }
    else { op[0] = m->spawn_pos[0]; op[1] = m->spawn_pos[1]; op[2] = m->spawn_pos[2]; }
    /* acceleration of the point of the PARENT body that coincides with this joint's origin */
    real r[3], t1[3], t2[3];
    v3sub(r, fk->o[j], op);
    v3cross(t1, alp, r);
    v3cross(t2, wp, r); v3cross(t2, wp, t2);
    for (int i = 0; i < 3; i++) ao[j][i] = aop[i] + t1[i] + t2[i];
    if (m->jtype[j] == 0) {
      /* prismatic: Coriolis 2 w_p x (a qd); origin moves with the joint (already in o_j) */
      real c[3]; v3cross(c, wp, fk->a[j]);
      for (int i = 0; i < 3; i++) { w[j][i] = wp[i]; al[j][i] = alp[i]; ao[j][i] += 2 * qd[j] * c[i]; }
    } else {
      real c[3]; v3cross(c, wp, fk->a[j]);
      for (int i = 0; i < 3; i++) { w[j][i] = wp[i] + qd[j] * fk->a[j][i]; al[j][i] = alp[i] + qd[j] * c[i]; }
    }
    /* body force / torque about COM */
    real c[3], I[9];
    body_world(m, fk, j, c, I);
    real rc[3], ac[3], u1[3], u2[3];
    v3sub(rc, c, fk->o[j]);
    v3cross(u1, al[j], rc);
    v3cross(u2, w[j], rc); v3cross(u2, w[j], u2);
    for (int i = 0; i < 3; i++) ac[i] = ao[j][i] + u1[i] + u2[i];
    real F[3], N[3], Iw[3], Ial[3];
    for (int i = 0; i < 3; i++) F[i] = m->mass[j] * ac[i];
    m3v(Iw, I, w[j]); m3v(Ial, I, al[j]);
    v3cross(N, w[j], Iw);
    for (int i = 0; i < 3; i++) N[i] += Ial[i];
    /* wrench about the joint origin */
    real rxF[3]; v3cross(rxF, rc, F);
    for (int i = 0; i < 3; i++) { f[j][i] = F[i]; n[j][i] = N[i] + rxF[i]; }
  }
  for (int j = NJ - 1; j >= 0; j--) {
    tau[j] = m->jtype[j] == 0 ? v3dot(fk->a[j], f[j]) : v3dot(fk->a[j], n[j]);
    int p = joint_parent(j);
    if (p >= 0) {
      real r[3], rxf[3];
      v3sub(r, fk->o[j], fk->o[p]);
      v3cross(rxf, r, f[j]);
      for (int i = 0; i < 3; i++) { f[p][i] += f[j][i]; n[p][i] += n[j][i] + rxf[i]; }
    }
  }
}

/* in-place Cholesky A = L L^T (lower), returns 0 on success */
static int cholesky(real* A, int n) {
  for (int i = 0; i < n; i++) {
    for (int j = 0; j <= i; j++) {
      real s = A[i * n + j];
      for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k];
      if (i == j) { if (s <= 0) return 1; A[i * n + i] = RSQRT(s); }
      else A[i * n + j] = s / A[j * n + j];
    }
  }
  return 0;
}
static void chol_solve(const real* L, int n, real* b) {
  for (int i = 0; i < n; i++) { real s = b[i]; for (int k = 0; k < i; k++) s -= L[i * n + k] * b[k]; b[i] = s / L[i * n + i]; }
  for (int i = n - 1; i >= 0; i--) { real s = b[i]; for (int k = i + 1; k < n; k++) s -= L[k * n + i] * b[k]; b[i] = s / L[i * n + i]; }
}

/* ------------------------------------------------------------------------------------------ narrowphase */
static inline void tangent_basis(const real* n, real* t1, real* t2) {
  real e[3] = {0, 0, 0};
  if (RFABS(n[0]) < (real)0.57735) e[0] = 1; else e[1] = 1;
  v3cross(t1, e, n);
  real l = v3norm(t1);
  for (int i = 0; i < 3; i++) t1[i] /= l;
  v3cross(t2, n, t1);
}

static inline void push_contact(OrcEnv* e, int type, int cap, int sample, const real* p, const real* n, real gap, real mu) {
  if (e->ncontact >= KMAX) return;
  OrcContact* c = &e->contact[e->ncontact++];
  v3cpy(c->p, p); v3cpy(c->n, n); c->gap = gap; c->mu = mu; c->type = type; c->cap = cap;
  c->key = type == 2 ? sample : (cap * 2 + type) * 2 + sample;
  c->finger = -1; c->side = 0;
}

/* Joint-limit rows of finger f (cfg.joint_limit_rows): every joint of the finger within the margin of a limit gets a one-row
 * speculative constraint, lower before upper, proximal to distal; they follow the finger's contacts in the list */
static inline void push_limit_rows(const Oracle* o, OrcEnv* e, int f) {
  const DexHandModel* m = &o->model;
  static const real zero3[3] = {0, 0, 0};
  for (int l = 0; l < 4; l++) {
    int j = 6 + 4 * f + l;
    real gap[2] = {e->q[j] - m->lo[j], m->hi[j] - e->q[j]};
    for (int sd = 0; sd < 2; sd++) {
      if (!(gap[sd] < o->cfg.joint_limit_margin) || e->ncontact >= KMAX) continue;
      push_contact(e, 3, l, sd, zero3, zero3, gap[sd], 0);
      OrcContact* c = &e->contact[e->ncontact - 1];
      c->finger = f; c->side = sd; c->key = 88 + (4 * f + l) * 2 + sd;
    }
  }
}

/* sphere (centre P in box frame, radius r) against the solid box of half extents h */
static inline int sphere_box(const real* P, real r, const real* h, real* nl, real* pl, real* gapraw) {
  real qv[3], d[3];
  for (int i = 0; i < 3; i++) { qv[i] = clampr(P[i], -h[i], h[i]); d[i] = P[i] - qv[i]; }
  real d2 = v3dot(d, d);
  if (d2 > (real)1e-12) {
    real dist = RSQRT(d2);
    for (int i = 0; i < 3; i++) { nl[i] = d[i] / dist; pl[i] = qv[i]; }
    *gapraw = dist - r;
  } else {
    int ax = 0; real best = h[0] - RFABS(P[0]);
    for (int i = 1; i < 3; i++) { real pen = h[i] - RFABS(P[i]); if (pen < best) { best = pen; ax = i; } }
    real sgn = P[ax] >= 0 ? (real)1 : (real)-1;
    for (int i = 0; i < 3; i++) { nl[i] = 0; pl[i] = P[i]; }
    nl[ax] = sgn; pl[ax] = sgn * h[ax];
    *gapraw = -best - r;
  }
  return 1;
}

/* derivative sign of dist^2(segment point, box) */
static inline real seg_box_dfdt(const real* a, const real* d, real t, const real* h) {
  real g = 0;
  for (int i = 0; i < 3; i++) { real P = a[i] + t * d[i]; real ex = P - clampr(P, -h[i], h[i]); g += ex * d[i]; }
  return g;
}

static void collide(const Oracle* o, OrcEnv* e, const FK* fk) {
  const DexSimConfig* cfg = &o->cfg;
  const DexHandModel* m = &o->model;
  const int limit_rows = cfg->joint_limit_rows;
  int nreal_v = 0, *nreal = &nreal_v;   /* contacts found for the current finger (before the list is cut at KMAX) */
  e->ncontact = 0;
  real co = cfg->contact_offset, rest = cfg->rest_offset;
  real zup[3] = {0, 0, 1};
  real Rb[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, hb[3] = {0, 0, 0};
  if (cfg->has_box) {
    q2mat(Rb, e->box_quat);
    hb[0] = hb[1] = hb[2] = (real)0.5 * cfg->box_size;
    real mu_bg = (real)0.5 * (e->box_mu + cfg->ground_friction);
    /* at most 4 box/ground contacts (a cube touches a plane with at most 4 corners; more only when it is buried), in
     * corner-index order */
    for (int i = 0; i < 8 && !cfg->box_fixed; i++) {   /* (a static box is not a dynamic body: no box/ground rows) */
      real l[3] = {(i & 1) ? hb[0] : -hb[0], (i & 2) ? hb[1] : -hb[1], (i & 4) ? hb[2] : -hb[2]}, p[3];
      m3v(p, Rb, l); v3add(p, p, e->box_pos);
      if (p[2] < co && e->ncontact < 4) push_contact(e, 2, -1, i, p, zup, p[2] - rest, mu_bg);
    }
  }
  real mu_hg = (real)0.5 * (m->hand_friction + cfg->ground_friction);
  real mu_hb = (real)0.5 * (m->hand_friction + e->box_mu);
  /* capsule order: finger 0..4, each distal -> middle -> proximal, then the palm */
  for (int k = 0; k < DEXSIM_NCAP; k++) {
    int c = k < 15 ? 3 + 3 * (k / 3) + (2 - k % 3) : k - 15;
    int j = m->cap_parent[c];
    real r = m->cap_r[c];
    real l0[3] = {m->cap_p0[c][0], m->cap_p0[c][1], m->cap_p0[c][2]}, l1[3] = {m->cap_p1[c][0], m->cap_p1[c][1], m->cap_p1[c][2]};
    real e0[3], e1[3];
    m3v(e0, fk->R[j], l0); v3add(e0, e0, fk->o[j]);
    m3v(e1, fk->R[j], l1); v3add(e1, e1, fk->o[j]);
    if (cfg->has_box) {
      real a[3], b[3], d[3], t0[3];
      v3sub(t0, e0, e->box_pos); m3tv(a, Rb, t0);
      v3sub(t0, e1, e->box_pos); m3tv(b, Rb, t0);
      v3sub(d, b, a);
      /* broadphase: bounding spheres */
      real mid[3] = {(real)0.5 * (a[0] + b[0]), (real)0.5 * (a[1] + b[1]), (real)0.5 * (a[2] + b[2])};
      real reach = (real)0.5 * v3norm(d) + r + (real)1.7320508 * hb[0] + co;
      if (v3dot(mid, mid) < reach * reach) {
        real lo = 0, hi = 1;
        for (int it = 0; it < DEXSIM_BISECT_ITERS; it++) {
          real mdl = (real)0.5 * (lo + hi);
          if (seg_box_dfdt(a, d, mdl, hb) > 0) hi = mdl; else lo = mdl;
        }
        real ts = (real)0.5 * (lo + hi);
        /* closest feature first (axis point nearest to the box, snapped to an end within 2 % of it), then the far end of
         * the axis: at most 2 contacts per capsule/box pair (round 1 sampled both ends + the closest point) */
        real tp = ts <= (real)0.02 ? (real)0 : (ts >= (real)0.98 ? (real)1 : ts);
        real tc[2] = {tp, tp < (real)0.5 ? (real)1 : (real)0};
        for (int s = 0; s < 2; s++) {
          real P[3] = {a[0] + tc[s] * d[0], a[1] + tc[s] * d[1], a[2] + tc[s] * d[2]}, nl[3], pl[3], gr;
          sphere_box(P, r, hb, nl, pl, &gr);
          if (gr < co) {
            real pw[3], nw[3];
            m3v(pw, Rb, pl); v3add(pw, pw, e->box_pos);
            m3v(nw, Rb, nl);
            push_contact(e, 1, c, s, pw, nw, gr - rest, mu_hb);
            (*nreal)++;
          }
        }
      }
    }
    const real* ends[2] = {e0, e1};
    for (int s = 0; s < 2; s++) {
      real zl = ends[s][2] - r;
      if (zl < co) { real p[3] = {ends[s][0], ends[s][1], zl}; push_contact(e, 0, c, s, p, zup, zl - rest, mu_hg); (*nreal)++; }
    }
    if (k < 15 && k % 3 == 2) {   /* behind the finger's three capsules: its joint-limit rows, if the finger touches anything */
      if (limit_rows && *nreal > 0) push_limit_rows(o, e, k / 3);
      *nreal = 0;
    }
  }
}

/* ------------------------------------------------------------------------------------------ one sub-step */
static void substep(const Oracle* o, OrcEnv* e, real h, int last) {
  const DexSimConfig* cfg = &o->cfg;
  const DexHandModel* m = &o->model;
  FK fk;
  fk_compute(m, e->q, &fk);
  static const real zero3[3] = {0, 0, 0};
  (void)zero3;
  real M[NJ][NJ], bias[NJ], g[3] = {cfg->gravity[0], cfg->gravity[1], cfg->gravity[2]};
  crba(m, &fk, M);
  rnea_bias(m, &fk, e->qd, g, bias);
  real Mh[NJ * NJ], tau[NJ];
  for (int i = 0; i < NJ; i++) {
    for (int j = 0; j < NJ; j++) Mh[i * NJ + j] = M[i][j];
    Mh[i * NJ + i] += m->armature[i] + h * (m->kd[i] + h * m->kp[i]);
    tau[i] = m->kp[i] * (e->targets[i] - e->q[i]) - (m->kd[i] + h * m->kp[i]) * e->qd[i] - bias[i];
  }
  if (cholesky(Mh, NJ)) { fprintf(stderr, "oracle: mass matrix not SPD\n"); return; }
  real v[NV];
  {
    real a[NJ];
    memcpy(a, tau, sizeof a);
    chol_solve(Mh, NJ, a);
    for (int i = 0; i < NJ; i++) v[i] = e->qd[i] + h * a[i];
  }
  real inv_m = 0, inv_I = 0;
  if (cfg->has_box && !cfg->box_fixed) {
    inv_m = 1 / e->box_mass;
    inv_I = 1 / (e->box_mass * cfg->box_size * cfg->box_size / 6); /* solid cube */
    for (int i = 0; i < 3; i++) { v[26 + i] = e->box_lin[i] + h * g[i]; v[29 + i] = e->box_ang[i]; }
  } else for (int i = 26; i < NV; i++) v[i] = 0;   /* no box, or the static box of the harness (fix_base_link): 1/m = 1/I = 0 */

  collide(o, e, &fk);
  int K = e->ncontact;
  /* rows: J (dense 32), Y = Mhat^-1 J^T, 1/D, bias */
  static __thread real J[KMAX * 3][NV], Y[KMAX * 3][NV];
  real Dinv[KMAX * 3], cbias[KMAX], lam[KMAX * 3];
  for (int k = 0; k < K; k++) {
    OrcContact* c = &e->contact[k];
    real dir[3][3];
    memset(dir, 0, sizeof dir);
    if (c->type != 3) { v3cpy(dir[0], c->n); tangent_basis(c->n, dir[1], dir[2]); }
    for (int r = 0; r < 3; r++) {
      real* Jr = J[3 * k + r];
      real* Yr = Y[3 * k + r];
      memset(Jr, 0, sizeof(real) * NV);
      if (c->type == 3) {   /* joint-limit row: J = +-e_j, one row (rows 1, 2 stay empty: 1/D = 0 below, mu = 0) */
        if (r == 0) Jr[6 + 4 * c->finger + c->cap] = c->side ? (real)-1 : (real)1;
      } else if (c->type != 2) {
        for (int j = m->cap_parent[c->cap]; j >= 0; j = joint_parent(j)) {
          real jv[3]; jac_col(m, &fk, j, c->p, jv);
          Jr[j] = v3dot(dir[r], jv);
        }
      }
      if (c->type == 1 || c->type == 2) {
        real sgn = c->type == 2 ? (real)1 : (real)-1, rb[3], rxd[3];
        v3sub(rb, c->p, e->box_pos);
        v3cross(rxd, rb, dir[r]);
        for (int i = 0; i < 3; i++) { Jr[26 + i] = sgn * dir[r][i]; Jr[29 + i] = sgn * rxd[i]; }
      }
      memcpy(Yr, Jr, sizeof(real) * NV);
      chol_solve(Mh, NJ, Yr);
      for (int i = 0; i < 3; i++) { Yr[26 + i] = Jr[26 + i] * inv_m; Yr[29 + i] = Jr[29 + i] * inv_I; }
      real D = 0;
      for (int i = 0; i < NV; i++) D += Jr[i] * Yr[i];
      Dinv[3 * k + r] = (c->type == 3 && r > 0) ? 0 : 1 / (D + (real)1e-9);
      lam[3 * k + r] = 0;
    }
    cbias[k] = c->gap > 0 ? c->gap / h : -minr(-c->gap * cfg->erp / h, cfg->max_depenetration_velocity);
  }
  if (K > 0) {
    /* ---- warm start: a contact that existed in the previous sub-step (same key) starts from the impulses it ended with;
     * the free velocity receives their effect */
    for (int k = 0; k < K; k++) {
      const OrcContact* c = &e->contact[k];
      int key = c->type == 2 ? 80 + k : c->key;
      if (e->wtag[key] != 8 * e->wgen + (c->type == 2 ? c->key : 0)) continue;
      for (int r = 0; r < 3; r++) {
        real l0 = e->wlam[key][r];
        lam[3 * k + r] = l0;
        for (int i = 0; i < NV; i++) v[i] += Y[3 * k + r][i] * l0;
      }
    }
    /* ---- block-parallel projected Gauss-Seidel with mass splitting (Tonge et al. 2012; PhysX's GPU solver is of this
     * family).  Blocks: the box/ground contacts together (block 12) and every hand contact on its own (hand contact j -> block
     * j; with more than 12 hand contacts: pairs of consecutive list entries, j -> block j / 2; blocks 0-11).  Within a sweep every block runs sequential projected Gauss-Seidel
     * over its own contacts, starting from the common velocities; a body that several blocks touch -- the hand base
     * (the 6 base DOFs at zero generalized finger momentum: T = [I; -G]), a finger's own DOFs, the box -- is shared out
     * equally: a block sees 1/n of its mass, n = the number of blocks touching it; after the sweep the velocity changes of
     * all blocks are added up.  Fixed point = the solution of the same LCP as plain Gauss-Seidel. */
    int blk[KMAX], fng[KMAX], cntb[13] = {0}, boxb[13] = {0}, fb[5][13];
    memset(fb, 0, sizeof fb);
    int jh = 0, nhand = 0;
    for (int k = 0; k < K; k++) nhand += e->contact[k].type != 2;
    const int run = nhand > 12 ? 2 : 1;   /* hand contacts per block: every contact its own block; pairs of consecutive list
                                            * entries when there are more than 12 (at most 12 hand blocks) */
    for (int k = 0; k < K; k++) {
      OrcContact* c = &e->contact[k];
      fng[k] = c->type == 2 ? -1 : (c->type == 3 ? c->finger : (m->cap_parent[c->cap] >= 6 ? (m->cap_parent[c->cap] - 6) / 4 : -1));
      blk[k] = c->type == 2 ? 12 : jh++ / run;
      cntb[blk[k]]++;
      if (c->type == 1 || c->type == 2) boxb[blk[k]] = 1;
      if (fng[k] >= 0) fb[fng[k]][blk[k]] = 1;
    }
    int nB = 0, nX = 0, nF[5];
    for (int b = 0; b < 12; b++) nB += cntb[b] > 0;
    for (int b = 0; b < 13; b++) nX += boxb[b];
    for (int f = 0; f < 5; f++) { nF[f] = 0; for (int b = 0; b < 13; b++) nF[f] += fb[f][b]; if (nF[f] < 1) nF[f] = 1; }
    if (nB < 1) nB = 1;
    if (nX < 1) nX = 1;
    /* T = [I; -G], G_f = Fhat_f^-1 C_f (Mh holds the Cholesky factor by now: take the blocks from M + the diagonal terms) */
    real Tm[NJ][6];
    memset(Tm, 0, sizeof Tm);
    for (int i = 0; i < 6; i++) Tm[i][i] = 1;
    for (int f = 0; f < 5 && nhand > 0; f++) {   /* (no hand contact: no row touches the hand, T is not needed) */
      real F[16], Cc[4][6];
      for (int i = 0; i < 4; i++) {
        int gi = 6 + 4 * f + i;
        for (int j = 0; j < 4; j++) F[i * 4 + j] = M[gi][6 + 4 * f + j];
        F[i * 4 + i] += m->armature[gi] + h * (m->kd[gi] + h * m->kp[gi]);
        for (int j = 0; j < 6; j++) Cc[i][j] = M[gi][j];
      }
      cholesky(F, 4);
      for (int j = 0; j < 6; j++) {
        real col[4] = {Cc[0][j], Cc[1][j], Cc[2][j], Cc[3][j]};
        chol_solve(F, 4, col);
        for (int i = 0; i < 4; i++) Tm[6 + 4 * f + i][j] = -col[i];
      }
    }
    /* response of a row inside its block: Y = T Y_B + P (P: the finger's own part) -> Ys = nB T Y_B + nF P, box part x nX */
    static __thread real Ys[KMAX * 3][NV];
    for (int row = 0; row < 3 * K; row++) {
      if (e->contact[row / 3].type == 2) for (int i = 0; i < NJ; i++) Ys[row][i] = 0;   /* box/ground rows: no hand part */
      else for (int i = 0; i < NJ; i++) {
        real t = 0;
        for (int j = 0; j < 6; j++) t += Tm[i][j] * Y[row][j];
        Ys[row][i] = i < 6 ? (real)nB * Y[row][i] : (real)nB * t + (real)nF[(i - 6) / 4] * (Y[row][i] - t);
      }
      for (int i = NJ; i < NV; i++) Ys[row][i] = (real)nX * Y[row][i];
      real D = 0;
      for (int i = 0; i < NV; i++) D += J[row][i] * Ys[row][i];
      Dinv[row] = (e->contact[row / 3].type == 3 && row % 3 > 0) ? 0 : 1 / (D + (real)1e-9);
    }
    for (int it = 0; it < cfg->num_position_iterations; it++) {
      real dsum[NV];
      memset(dsum, 0, sizeof dsum);
      for (int b = 0; b < 13; b++) {
        if (!cntb[b]) continue;
        real vb[NV], db[NV];
        memcpy(vb, v, sizeof vb);
        memset(db, 0, sizeof db);
        for (int k = 0; k < K; k++) {
          if (blk[k] != b) continue;
          real mu = e->contact[k].mu;
          for (int r = 0; r < 3; r++) {
            int row = 3 * k + r;
            real vr = 0;
            for (int i = 0; i < NV; i++) vr += J[row][i] * vb[i];
            real nl;
            if (r == 0) nl = maxr(0, lam[row] - (vr + cbias[k]) * Dinv[row]);
            else { real lim = mu * lam[3 * k]; nl = clampr(lam[row] - vr * Dinv[row], -lim, lim); }
            real dl = nl - lam[row];
            lam[row] = nl;
            for (int i = 0; i < NV; i++) { vb[i] += Ys[row][i] * dl; db[i] += Y[row][i] * dl; }
          }
        }
        for (int i = 0; i < NV; i++) dsum[i] += db[i];
      }
      for (int i = 0; i < NV; i++) v[i] += dsum[i];
    }
  }
  {
    e->wgen = DEXSIM_WGEN_NEXT(e->wgen);
    for (int k = 0; k < K; k++) {
      const OrcContact* c = &e->contact[k];
      int key = c->type == 2 ? 80 + k : c->key;
      e->wtag[key] = 8 * e->wgen + (c->type == 2 ? c->key : 0);
      for (int r = 0; r < 3; r++) e->wlam[key][r] = lam[3 * k + r];
    }
  }
  /* net contact force per body, last sub-step only (contact_collection: 1 = CC_LAST_SUBSTEP) */
  if (last) {
    memset(e->cforce, 0, sizeof e->cforce);
    for (int k = 0; k < K; k++) {
      OrcContact* c = &e->contact[k];
      real dir[3][3], F[3] = {0, 0, 0};
      if (c->type == 3) continue;   /* a joint-limit reaction is a joint torque, not a force on a body */
      v3cpy(dir[0], c->n);
      tangent_basis(c->n, dir[1], dir[2]);
      for (int r = 0; r < 3; r++) v3axpy(F, lam[3 * k + r] / h, dir[r]);
      if (c->type != 2) { int s = m->cap_fslot[c->cap]; for (int i = 0; i < 3; i++) e->cforce[s][i] += F[i]; }
      if (c->type == 2) for (int i = 0; i < 3; i++) e->cforce[DEXSIM_FSLOT_BOX][i] += F[i];
      if (c->type == 1) for (int i = 0; i < 3; i++) e->cforce[DEXSIM_FSLOT_BOX][i] -= F[i];
    }
  }
  /* integrate (semi-implicit Euler) + joint-limit clamp */
  for (int i = 0; i < NJ; i++) {
    real qn = e->q[i] + h * v[i], vn = v[i];
    if (qn < m->lo[i]) { qn = m->lo[i]; vn = maxr(vn, 0); }
    if (qn > m->hi[i]) { qn = m->hi[i]; vn = minr(vn, 0); }
    e->q[i] = qn; e->qd[i] = vn;
  }
  if (cfg->has_box && !cfg->box_fixed) {
    for (int i = 0; i < 3; i++) { e->box_lin[i] = v[26 + i]; e->box_ang[i] = v[29 + i]; e->box_pos[i] += h * v[26 + i]; }
    real wq[4] = {e->box_ang[0], e->box_ang[1], e->box_ang[2], 0}, dq[4];
    qmul(dq, wq, e->box_quat);
    real qn[4], l = 0;
    for (int i = 0; i < 4; i++) { qn[i] = e->box_quat[i] + (real)0.5 * h * dq[i]; l += qn[i] * qn[i]; }
    l = RSQRT(l);
    for (int i = 0; i < 4; i++) e->box_quat[i] = qn[i] / l;
  }
}

/* compact L1 state: site poses, hand-base twist (gym.refresh_rigid_body_state_tensor for the rows the
 * path reads: observation_encoder.py:645-647,849-851,874-876; reward_calculator.py:86-91) */
static void publish(const Oracle* o, OrcEnv* e) {
  const DexHandModel* m = &o->model;
  FK fk;
  fk_compute(m, e->q, &fk);
  for (int s = 0; s < DEXSIM_NSITE; s++) {
    int j = m->site_parent[s];
    real lp[3] = {m->site_p[s][0], m->site_p[s][1], m->site_p[s][2]}, t[3];
    m3v(t, fk.R[j], lp);
    v3add(e->site_pose[s], fk.o[j], t);
    real sq[4] = {m->site_q[s][0], m->site_q[s][1], m->site_q[s][2], m->site_q[s][3]};
    qmul(&e->site_pose[s][3], fk.q[j], sq);
  }
  /* twist of right_hand_base (site 0): linear velocity of its origin + angular velocity */
  real lin[3] = {0, 0, 0}, ang[3] = {0, 0, 0};
  for (int j = 0; j <= 5; j++) {
    real jv[3]; jac_col(m, &fk, j, e->site_pose[0], jv);
    v3axpy(lin, e->qd[j], jv);
    if (m->jtype[j] == 1) v3axpy(ang, e->qd[j], fk.a[j]);
  }
  for (int i = 0; i < 3; i++) { e->hand_vel[i] = lin[i]; e->hand_vel[3 + i] = ang[i]; }
}

static void physics_step_env(const Oracle* o, OrcEnv* e) {
  real h = o->cfg.dt / (real)o->cfg.substeps;
  for (int s = 0; s < o->cfg.substeps; s++) substep(o, e, h, s == o->cfg.substeps - 1);
  publish(o, e);
}

/* ------------------------------------------------------------------------------------------ L2: actions */
static const int COUPLING[19][2] = { /* (finger control, dof)  constants.py:71-88 in DOF-name order */
  {0, 6}, {1, 7}, {2, 8}, {2, 9}, {3, 10}, {3, 18}, {3, 22}, {4, 11}, {5, 12}, {5, 13},
  {6, 15}, {7, 16}, {7, 17}, {8, 19}, {9, 20}, {9, 21}, {10, 23}, {11, 24}, {11, 25}};
static inline real coupling_scale(int dof) { return dof == 22 ? (real)2 : (real)1; }
#define MIDDLE_SPREAD_DOF 14
/* HardwareMapping enum order -> primary DOF (hand_initializer.py:20-38, observation_encoder.py:496-574) */
static const int ACTIVE_FINGER_DOF[12] = {8, 7, 6, 10, 12, 11, 16, 15, 20, 19, 24, 23};

static void process_actions_env(const Oracle* o, OrcEnv* e, const float* act, int zero_targets) {
  const DexSimConfig* c = &o->cfg;
  int na = c->num_actions;
  for (int i = 0; i < 18; i++) e->actions[i] = i < na ? (real)act[i] : 0;
  if (zero_targets) { /* action_processor.py:305-318 */
    for (int i = 0; i < NJ; i++) e->targets[i] = 0;
  } else {
    real raw[18], nxt[18];
    const real* prev = e->active_prev_targets;
    for (int i = 0; i < 18; i++) raw[i] = e->active_rule_targets[i];
    int fs = c->policy_controls_base ? 6 : 0;
    if (c->control_mode == DEXSIM_MODE_POSITION_DELTA) { /* default_rules.py:80-112 */
      if (c->policy_controls_base) for (int i = 0; i < 6; i++) raw[i] = prev[i] + e->actions[i] * c->max_deltas[i];
      if (c->policy_controls_fingers) for (int i = 0; i < 12; i++) raw[6 + i] = prev[6 + i] + e->actions[fs + i] * c->max_deltas[6 + i];
      for (int i = 0; i < 18; i++) raw[i] = clampr(raw[i], c->active_lower[i], c->active_upper[i]);
    } else { /* default_rules.py:33-64, scaling.py:28-47 */
      if (c->policy_controls_base) for (int i = 0; i < 6; i++)
        raw[i] = (e->actions[i] + (real)1.0) * (real)0.5 * (c->active_upper[i] - c->active_lower[i]) + c->active_lower[i];
      if (c->policy_controls_fingers) for (int i = 0; i < 12; i++)
        raw[6 + i] = (e->actions[fs + i] + (real)1.0) * (real)0.5 * (c->active_upper[6 + i] - c->active_lower[6 + i]) + c->active_lower[6 + i];
    }
    for (int i = 0; i < 18; i++) { /* rules.py:141-190 */
      real d = clampr(raw[i] - prev[i], -c->max_deltas[i], c->max_deltas[i]);
      nxt[i] = clampr(prev[i] + d, c->active_lower[i], c->active_upper[i]);
    }
    for (int i = 0; i < NJ; i++) e->targets[i] = 0; /* action_processor.py:571-614 */
    for (int i = 0; i < 6; i++) e->targets[i] = nxt[i];
    for (int k = 0; k < 19; k++) e->targets[COUPLING[k][1]] = nxt[6 + COUPLING[k][0]] * coupling_scale(COUPLING[k][1]);
    e->targets[MIDDLE_SPREAD_DOF] = 0;
    for (int i = 0; i < 18; i++) e->active_prev_targets[i] = nxt[i];
  }
  for (int i = 0; i < 18; i++) e->prev_actions[i] = e->actions[i]; /* observation_encoder.py:287-296 */
}

/* inverse coupling with the CPU last-writer-wins scatter (action_processor.py:616-666) */
static void extract_active_targets(const real* full, real* active) {
  for (int i = 0; i < 18; i++) active[i] = 0;
  for (int i = 0; i < 6; i++) active[i] = full[i];
  for (int dof = 6; dof < NJ; dof++) {
    int ctl = -1;
    for (int k = 0; k < 19; k++) if (COUPLING[k][1] == dof) ctl = COUPLING[k][0];
    if (ctl >= 0) active[6 + ctl] = full[dof] / coupling_scale(dof);
  }
}

/* ------------------------------------------------------------------------------------------ L2: observations */
enum { O_base_dof_pos = 0, O_base_dof_vel = 6, O_active_finger_dof_pos = 12, O_active_finger_dof_vel = 24,
       O_all_finger_dof_pos = 36, O_all_finger_dof_vel = 56, O_hand_pose = 76, O_hand_pose_arr_aligned = 83,
       O_contact_forces = 90, O_prev_actions = 105, O_active_prev_targets = 123, O_base_dof_target = 141,
       O_active_finger_dof_target = 147, O_all_finger_dof_target = 159, O_contact_force_magnitude = 179,
       O_contact_binary = 184, O_contact_duration = 189, O_fingertip_poses_world = 194,
       O_fingertip_poses_hand = 229, O_fingerpad_poses_world = 264, O_fingerpad_poses_hand = 299,
       O_episode_time = 334, O_active_rule_targets = 335, O_object_pos = 353, O_object_vel = 356,
       O_finger_to_object_distances = 359, O_avg_finger_to_object_distance = 364,
       O_finger_to_object_height_diff = 365, O_avg_finger_to_object_height_diff = 370,
       O_hand_to_object_distance = 371, O_fingerpad_distances = 372, O_first_three_fingerpad_centroid = 382,
       O_thumb_contact = 385, O_other_fingers_contact = 386, O_grasp_state = 387, O_grasp_duration = 388,
       O_current_stage = 389, O_time_in_stage = 390, O_stage_progress = 391 };

static inline const real* finger_cforce(const OrcEnv* e, int f) { return e->cf5[f]; }
static inline void refresh_cf5(OrcEnv* e) { /* r_f_link{f+1}_4 = force slot 3f+2 */
  for (int f = 0; f < 5; f++) for (int i = 0; i < 3; i++) e->cf5[f][i] = e->cforce[3 * f + 2][i];
}

/* blind_grasping_task.py:773-833 */
static void detect_finger_box_contacts(const Oracle* o, const OrcEnv* e, const real* ob, int* out) {
  const DexSimConfig* c = &o->cfg;
  real bm = v3norm(e->cforce[DEXSIM_FSLOT_BOX]);
  int box_has = bm > c->contact_binary_threshold;
  real prox = (real)(sqrt(3.0) * (double)c->box_size / 2 * 1.2);
  for (int f = 0; f < 5; f++) {
    real d[3]; v3sub(d, e->site_pose[6 + f], e->box_pos);
    int near = v3norm(d) < prox;
    out[f] = (ob[O_contact_binary + f] != 0) && box_has && near;
  }
}

static void compute_observations_env(const Oracle* o, OrcEnv* e) {
  const DexSimConfig* c = &o->cfg;
  real* ob = e->obs_all;
  real cdt = c->control_dt;
  /* manual finite-difference velocity (observation_encoder.py:298-321) */
  real vel[NJ];
  for (int i = 0; i < NJ; i++) { vel[i] = (e->q[i] - e->prev_dof_pos[i]) / cdt; e->prev_dof_pos[i] = e->q[i]; }
  for (int i = 0; i < 6; i++) { ob[O_base_dof_pos + i] = e->q[i]; ob[O_base_dof_vel + i] = vel[i]; ob[O_base_dof_target + i] = e->targets[i]; }
  for (int i = 0; i < 12; i++) {
    int d = ACTIVE_FINGER_DOF[i];
    ob[O_active_finger_dof_pos + i] = e->q[d]; ob[O_active_finger_dof_vel + i] = vel[d]; ob[O_active_finger_dof_target + i] = e->targets[d];
  }
  for (int i = 0; i < 20; i++) { ob[O_all_finger_dof_pos + i] = e->q[6 + i]; ob[O_all_finger_dof_vel + i] = vel[6 + i]; ob[O_all_finger_dof_target + i] = e->targets[6 + i]; }
  const real* hp = e->site_pose[0];
  for (int i = 0; i < 7; i++) ob[O_hand_pose + i] = hp[i];
  { /* :1483-1521  q_hand (x) conj([0, sqrt.5, 0, sqrt.5]) */
    real sh = (real)0.7071067811865476, inv[4] = {0, -sh, 0, sh};
    for (int i = 0; i < 3; i++) ob[O_hand_pose_arr_aligned + i] = hp[i];
    qmul(&ob[O_hand_pose_arr_aligned + 3], hp + 3, inv);
  }
  int cb[5];
  for (int f = 0; f < 5; f++) {
    const real* F = finger_cforce(e, f);
    for (int i = 0; i < 3; i++) ob[O_contact_forces + 3 * f + i] = F[i];
    real mag = v3norm(F);
    ob[O_contact_force_magnitude + f] = mag;
    cb[f] = mag > c->contact_binary_threshold;
    ob[O_contact_binary + f] = cb[f] ? (real)1 : (real)0;
    /* contact duration tracking (:323-361) */
    real cur = cb[f] ? (real)1 : (real)0;
    int started = (cur == 1) && (e->prev_contact_binary[f] == 0);
    e->contact_duration_steps[f] = started ? (real)1 : (cur == 1 ? e->contact_duration_steps[f] + 1 : (real)0);
    e->prev_contact_binary[f] = cur;
    ob[O_contact_duration + f] = e->contact_duration_steps[f] * cdt;
  }
  for (int i = 0; i < 18; i++) {
    ob[O_prev_actions + i] = e->prev_actions[i];
    ob[O_active_prev_targets + i] = e->active_prev_targets[i];
  }
  for (int f = 0; f < 5; f++) {
    for (int s = 0; s < 2; s++) { /* s=0 tips (sites 1..5), s=1 pads (sites 6..10) */
      const real* pw = e->site_pose[(s ? 6 : 1) + f];
      real* w = &ob[(s ? O_fingerpad_poses_world : O_fingertip_poses_world) + 7 * f];
      real* hnd = &ob[(s ? O_fingerpad_poses_hand : O_fingertip_poses_hand) + 7 * f];
      for (int i = 0; i < 7; i++) w[i] = pw[i];
      real rel[3], qc[4];
      v3sub(rel, pw, hp);
      qrot_inv(hnd, hp + 3, rel);            /* coordinate_transforms.py:17-35 */
      qconj(qc, hp + 3);
      qmul(hnd + 3, qc, pw + 3);             /* observation_encoder.py:971-974 */
    }
  }
  ob[O_episode_time] = (real)e->episode_step * cdt; /* dexhand_base.py:702-712 */

  if (c->task != DEXSIM_TASK_BLIND_GRASPING) return;
  /* ---- BlindGraspingTask.get_task_observations (blind_grasping_task.py:549-653) */
  for (int i = 0; i < 3; i++) { ob[O_object_pos + i] = e->box_pos[i]; ob[O_object_vel + i] = e->box_lin[i]; }
  real sumd = 0, sumh = 0;
  for (int f = 0; f < 5; f++) {
    real d[3]; v3sub(d, e->site_pose[6 + f], e->box_pos);
    real dist = v3norm(d), hd = RFABS(e->site_pose[6 + f][2] - e->box_pos[2]);
    ob[O_finger_to_object_distances + f] = dist; sumd += dist;
    ob[O_finger_to_object_height_diff + f] = hd; sumh += hd;
  }
  ob[O_avg_finger_to_object_distance] = sumd / 5;
  ob[O_avg_finger_to_object_height_diff] = sumh / 5;
  { real d[3]; v3sub(d, hp, e->box_pos); ob[O_hand_to_object_distance] = v3norm(d); }
  { int k = 0;
    for (int a = 0; a < 5; a++) for (int b = a + 1; b < 5; b++) {
      real d[3]; v3sub(d, e->site_pose[6 + a], e->site_pose[6 + b]); ob[O_fingerpad_distances + k++] = v3norm(d);
    } }
  for (int i = 0; i < 3; i++)
    ob[O_first_three_fingerpad_centroid + i] = (e->site_pose[6][i] + e->site_pose[7][i] + e->site_pose[8][i]) / 3;
  int fb[5];
  detect_finger_box_contacts(o, e, ob, fb);
  int thumb = fb[0], others = fb[1] || fb[2] || fb[3] || fb[4], grasp = thumb && others;
  ob[O_thumb_contact] = (real)thumb; ob[O_other_fingers_contact] = (real)others; ob[O_grasp_state] = (real)grasp;
  int nf = fb[0] + fb[1] + fb[2] + fb[3] + fb[4];
  int cond = (e->box_pos[2] > c->height_threshold) && (nf >= c->min_fingers_for_grasp);
  e->success_duration_steps = cond ? e->success_duration_steps + 1 : 0;
  e->success_conditions_met = cond;
  ob[O_grasp_duration] = (real)e->success_duration_steps * cdt;
  /* ---- stage FSM (:835-978) */
  e->time_in_stage += cdt;
  int tao = cb[0] && (cb[1] || cb[2] || cb[3] || cb[4]);
  if (e->current_stage == 2) e->stage_contact_duration = tao ? e->stage_contact_duration + cdt : (real)0;
  e->just2 = 0; e->just3 = 0;
  int s1done = (e->current_stage == 1) && (e->time_in_stage >= c->stage1_duration);
  int s2succ = (e->current_stage == 2) && (e->stage_contact_duration >= c->stage2_contact_success_threshold);
  int s2tout = (e->current_stage == 2) && (e->time_in_stage >= c->stage2_duration);
  int s2done = s2succ || s2tout;
  if (s1done) e->just2 = 1;
  if (s2done) e->just3 = 1;
  if (s1done) e->current_stage = 2;
  if (s2done) e->current_stage = 3;
  if (s1done || s2done) { e->time_in_stage = 0; e->stage_contact_duration = 0; }
  ob[O_current_stage] = (real)e->current_stage;
  ob[O_time_in_stage] = e->time_in_stage;
  real prog = 0;
  if (e->current_stage == 1) prog = clampr(e->time_in_stage / c->stage1_duration, 0, 1);
  if (e->current_stage == 2) prog = clampr(e->time_in_stage / c->stage2_duration, 0, 1);
  if (e->current_stage == 3) prog = 1;
  ob[O_stage_progress] = prog;
}

static void concat_observations_env(const Oracle* o, OrcEnv* e, int idx) {
  const DexSimConfig* c = &o->cfg;
  /* pre-action rule = identity clone (rules.py:78-95) */
  for (int i = 0; i < 18; i++) { e->active_rule_targets[i] = e->active_prev_targets[i]; e->obs_all[O_active_rule_targets + i] = e->active_rule_targets[i]; }
  float* dst = o->obs_buf + (size_t)idx * c->num_obs;
  int k = 0;
  for (int s = 0; s < c->n_obs_seg; s++)
    for (int i = 0; i < c->obs_seg_len[s]; i++) dst[k++] = (float)e->obs_all[c->obs_seg_off[s] + i];
}

/* ------------------------------------------------------------------------------------------ L2: termination + reward */
static int pregrasp_quality(const Oracle* o, const OrcEnv* e) { /* blind_grasping_task.py:1334-1364 */
  const DexSimConfig* c = &o->cfg;
  const real* ob = e->obs_all;
  int height_ok = 1;
  for (int f = 0; f < 3; f++) if (!(RFABS(e->site_pose[6 + f][2] - e->box_pos[2]) <= c->height_tolerance)) height_ok = 0;
  real d[3]; v3sub(d, &ob[O_first_three_fingerpad_centroid], e->box_pos);
  int centroid_ok = v3norm(d) <= c->centroid_tolerance;
  real dr[3]; v3sub(dr, e->box_pos, e->initial_box_pos);
  int stab_ok = (v3norm(dr) <= c->position_drift_tolerance) && (v3norm(e->box_lin) <= c->velocity_tolerance);
  return height_ok && centroid_ok && stab_ok;
}

static void evaluate_termination_env(const Oracle* o, OrcEnv* e) {
  const DexSimConfig* c = &o->cfg;
  const real* ob = e->obs_all;
  int crit_f[DEXSIM_NUM_FAIL] = {0}, crit_s[DEXSIM_NUM_SUCC] = {0};
  if (c->height_safety_enabled) { /* step_processor.py:141-164 */
    real minz = e->site_pose[1][2];
    for (int f = 1; f < 5; f++) minz = minr(minz, e->site_pose[1 + f][2]);
    crit_f[DEXSIM_FAIL_hitting_ground] = (e->site_pose[0][2] < c->handbase_threshold) || (minz < c->fingertip_threshold);
  }
  if (c->task == DEXSIM_TASK_BLIND_GRASPING) {
    crit_s[0] = e->success_duration_steps >= c->contact_duration_threshold_steps; /* :1210-1236 */
    crit_f[DEXSIM_FAIL_box_too_far] = ob[O_hand_to_object_distance] > c->max_box_distance;
    int grasp = ob[O_grasp_state] != 0;
    crit_f[DEXSIM_FAIL_stage1_pregrasp_failed] = e->just2 && !pregrasp_quality(o, e);
    crit_f[DEXSIM_FAIL_stage2_contact_failed] = e->just3 && !grasp;
    crit_f[DEXSIM_FAIL_stage3_grasp_lost] = (e->current_stage == 3) && !grasp;
  }
  for (int i = 0; i < DEXSIM_NUM_SUCC; i++) e->crit_success[i] = crit_s[i];
  for (int i = 0; i < DEXSIM_NUM_FAIL; i++) e->crit_failure[i] = crit_f[i];
  /* TerminationManager.evaluate (termination_manager.py:120-278) */
  int es = 0, ef = 0;
  for (int i = 0; i < DEXSIM_NUM_SUCC; i++) if (c->active_success_mask >> i & 1) {
    e->success_reason[i] |= (!es && crit_s[i]); es |= crit_s[i];
  }
  for (int i = 0; i < DEXSIM_NUM_FAIL; i++) if (c->active_failure_mask >> i & 1) {
    e->failure_reason[i] |= (!ef && crit_f[i]); ef |= crit_f[i];
  }
  int timeout = e->episode_step >= c->episode_length - 1;
  e->episode_success = es; e->episode_failure = ef;
  int should_reset = es || ef || timeout;
  e->term_success = es && should_reset;
  e->term_failure = ef && !es && should_reset;
  e->term_timeout = timeout && !es && !ef && should_reset;
  e->reset_flag = should_reset;
}

static void compute_rewards_env(const Oracle* o, OrcEnv* e, int first_call) {
  const DexSimConfig* c = &o->cfg;
  const DexHandModel* m = &o->model;
  const real* ob = e->obs_all;
  real r[DEXSIM_NUM_REWARD_TERMS];
  for (int i = 0; i < DEXSIM_NUM_REWARD_TERMS; i++) r[i] = 0;
  /* ---- common terms (reward_calculator.py:66-221); engine dof_vel, NOT the manual velocity */
  const real* hv = e->hand_vel; const real* hw = e->hand_vel + 3;
  r[DEXSIM_REW_alive] = 1;
  real minz = e->site_pose[1][2];
  for (int f = 1; f < 5; f++) minz = minr(minz, e->site_pose[1 + f][2]);
  r[DEXSIM_REW_height_safety] = clampr((real)1.0 - REXP(-(minz - (real)0.02) * 20), 0, 1);
  real fv2 = 0;
  for (int i = 0; i < 20; i++) fv2 += e->qd[6 + i] * e->qd[6 + i];
  r[DEXSIM_REW_finger_velocity] = REXP((real)-0.1 * RSQRT(fv2));
  r[DEXSIM_REW_hand_velocity] = REXP((real)-0.2 * v3norm(hv));
  r[DEXSIM_REW_hand_angular_velocity] = REXP((real)-0.2 * v3norm(hw));
  real pen = 0;
  for (int i = 0; i < 20; i++) {
    real lo = m->lo[6 + i], hi = m->hi[6 + i];
    real nrm = (real)2.0 * (e->q[6 + i] - lo) / (hi - lo) - (real)1.0;
    pen += clampr(RFABS(nrm) - (real)0.8, 0, 1);
  }
  r[DEXSIM_REW_joint_limit] = (real)1.0 - pen / 20;
  int contacts[5];
  for (int f = 0; f < 5; f++) contacts[f] = v3norm(&ob[O_contact_forces + 3 * f]) > (real)0.1;
  if (first_call) {
    for (int i = 0; i < 20; i++) e->prev_finger_dof_vel[i] = e->qd[6 + i];
    for (int i = 0; i < 3; i++) { e->prev_hand_vel[i] = hv[i]; e->prev_hand_ang_vel[i] = hw[i]; }
    for (int f = 0; f < 5; f++) e->prev_contacts[f] = contacts[f];
  }
  real fa2 = 0, ha[3], wa[3];
  for (int i = 0; i < 20; i++) { real d = e->qd[6 + i] - e->prev_finger_dof_vel[i]; fa2 += d * d; }
  v3sub(ha, hv, e->prev_hand_vel); v3sub(wa, hw, e->prev_hand_ang_vel);
  r[DEXSIM_REW_finger_acceleration] = REXP((real)-2.0 * RSQRT(fa2));
  r[DEXSIM_REW_hand_acceleration] = REXP((real)-0.5 * v3norm(ha));
  r[DEXSIM_REW_hand_angular_acceleration] = REXP((real)-0.5 * v3norm(wa));
  real changes = 0;
  for (int f = 0; f < 5; f++) changes += (contacts[f] != e->prev_contacts[f]) ? (real)1 : (real)0;
  r[DEXSIM_REW_contact_stability] = REXP(-changes);
  for (int i = 0; i < 20; i++) e->prev_finger_dof_vel[i] = e->qd[6 + i];
  for (int i = 0; i < 3; i++) { e->prev_hand_vel[i] = hv[i]; e->prev_hand_ang_vel[i] = hw[i]; }
  for (int f = 0; f < 5; f++) e->prev_contacts[f] = contacts[f];

  if (c->task == DEXSIM_TASK_BLIND_GRASPING) { /* blind_grasping_task.py:980-1208 */
    real s1 = e->current_stage == 1, s2 = e->current_stage == 2, s3 = e->current_stage == 3;
    r[DEXSIM_REW_s1_height_alignment] = REXP(-c->height_alignment_decay * ob[O_avg_finger_to_object_height_diff]) * s1;
    real d[3]; v3sub(d, &ob[O_first_three_fingerpad_centroid], e->box_pos);
    r[DEXSIM_REW_s1_centroid_positioning] = REXP(-c->centroid_positioning_decay * v3norm(d)) * s1;
    real dr[3]; v3sub(dr, e->box_pos, e->initial_box_pos);
    r[DEXSIM_REW_s1_object_stability] = REXP(-c->object_stability_decay * (v3norm(dr) + v3norm(e->box_lin))) * s1;
    real z0 = e->site_pose[6][2], z1 = e->site_pose[7][2], z2 = e->site_pose[8][2], mean = (z0 + z1 + z2) / 3;
    real var = ((z0 - mean) * (z0 - mean) + (z1 - mean) * (z1 - mean) + (z2 - mean) * (z2 - mean)) / 2; /* unbiased */
    r[DEXSIM_REW_s1_finger_height_consistency] = REXP(-c->first_three_height_consistency_decay * var) * s1;
    r[DEXSIM_REW_s1_thumb_rotation] = REXP((real)-5.0 * RFABS(ob[O_all_finger_dof_pos] - (real)(3.14159265358979323846 / 2))) * s1;
    r[DEXSIM_REW_s2_thumb_contact] = ob[O_thumb_contact] * s2;
    r[DEXSIM_REW_s2_other_fingers_contact] = ob[O_other_fingers_contact] * s2;
    r[DEXSIM_REW_s2_grasp_achievement] = ob[O_grasp_state] * s2;
    real mind = 0; /* closest fingertip to the box centre (:1178-1192) */
    for (int f = 0; f < 5; f++) { real t[3]; v3sub(t, e->site_pose[1 + f], e->box_pos); real n = v3norm(t); mind = f == 0 ? n : minr(mind, n); }
    real half = c->box_size / (real)2.0;
    real safe = maxr(mind, half * c->proximity_min_distance_factor);
    r[DEXSIM_REW_s2_fingerpad_proximity] = REXP(-c->fingerpad_proximity_decay * safe) * s2;
    real bv2 = 0; for (int i = 0; i < 6; i++) bv2 += ob[O_base_dof_vel + i] * ob[O_base_dof_vel + i];
    r[DEXSIM_REW_s2_base_stability] = REXP(-c->base_stability_decay * RSQRT(bv2)) * s2;
    r[DEXSIM_REW_s3_object_height] = clampr((e->box_pos[2] - c->box_z) / (c->height_threshold - c->box_z), 0, 1) * s3;
    r[DEXSIM_REW_s3_grasp_maintenance] = ob[O_grasp_state] * s3;
    r[DEXSIM_REW_s3_grasp_duration] = clampr(ob[O_grasp_duration] / c->contact_duration_threshold_s, 0, 1) * s3;
    r[DEXSIM_REW_s1_completion] = (e->just2 && !e->crit_failure[DEXSIM_FAIL_stage1_pregrasp_failed]) ? (real)1 : (real)0;
    r[DEXSIM_REW_s2_completion] = (e->just3 && !e->crit_failure[DEXSIM_FAIL_stage2_contact_failed]) ? (real)1 : (real)0;
    r[DEXSIM_REW_penetration_penalty] = maxr(half * c->geometric_penetration_factor - mind, 0) * c->penetration_depth_scale;
  }
  /* compute_total_reward (reward_calculator.py:223-275) + _add_termination_rewards (step_processor.py:204-219) */
  real total = 0;
  for (int i = 0; i < DEXSIM_NUM_REWARD_TERMS; i++) {
    real w = c->reward_weight[i], wr = r[i] * w;
    e->rew_comp[i] = r[i];
    e->rew_comp[DEXSIM_REWROW_WEIGHTED + i] = wr;
    if (w != 0) total += wr;
  }
  e->rew_comp[DEXSIM_REWROW_TOTAL] = total;
  real ts = e->term_success ? c->success_reward : 0, tf = e->term_failure ? -c->failure_penalty : 0, tt = e->term_timeout ? -c->timeout_penalty : 0;
  e->rew_comp[DEXSIM_REWROW_TERM_RAW + 0] = (real)e->term_success;
  e->rew_comp[DEXSIM_REWROW_TERM_RAW + 1] = (real)e->term_failure;
  e->rew_comp[DEXSIM_REWROW_TERM_RAW + 2] = (real)e->term_timeout;
  e->rew_comp[DEXSIM_REWROW_TERM_W + 0] = ts; e->rew_comp[DEXSIM_REWROW_TERM_W + 1] = tf; e->rew_comp[DEXSIM_REWROW_TERM_W + 2] = tt;
  e->rew = total + ts + tf + tt;
}

/* ------------------------------------------------------------------------------------------ resets */
static void reset_env(const Oracle* o, OrcEnv* e, int idx) {
  const DexSimConfig* c = &o->cfg;
  /* TerminationManager.reset_tracking (termination_manager.py:341-357) */
  e->episode_success = e->episode_failure = 0;
  for (int i = 0; i < DEXSIM_NUM_SUCC; i++) e->success_reason[i] = 0;
  for (int i = 0; i < DEXSIM_NUM_FAIL; i++) e->failure_reason[i] = 0;
  /* ResetManager.reset_idx (reset_manager.py:114-124) */
  e->episode_step = 0;
  for (int i = 0; i < NJ; i++) { e->q[i] = 0; e->qd[i] = 0; }
  if (c->task == DEXSIM_TASK_BLIND_GRASPING) { /* blind_grasping_task.py:433-547 */
    real u[DEXSIM_NRESET_SAMPLES];
    if (o->reset_samples) for (int i = 0; i < DEXSIM_NRESET_SAMPLES; i++) u[i] = o->reset_samples[(size_t)idx * DEXSIM_NRESET_SAMPLES + i];
    else for (int b = 0; b < 8; b++) {
      uint32_t x[4];
      philox4x32(x, (uint32_t)idx, (uint32_t)e->reset_count, (uint32_t)b, 0, c->seed, 0x5eedu);
      for (int i = 0; i < 4; i++) if (4 * b + i < DEXSIM_NRESET_SAMPLES) u[4 * b + i] = u01(x[i]);
    }
    real x = (u[0] * 2 - 1) * c->box_xy_range, y = (u[1] * 2 - 1) * c->box_xy_range;
    real yaw = (u[2] * 2 - 1) * (real)3.14159265358979323846;
    if (!c->box_fixed) {   /* (a static box stays where it was created) */
      v3set(e->box_pos, x, y, c->box_z);
      v3set(e->initial_box_pos, x, y, c->box_z);
      e->box_quat[0] = 0; e->box_quat[1] = 0; e->box_quat[2] = RSIN(yaw / 2); e->box_quat[3] = RCOS(yaw / 2);
    }
    for (int i = 0; i < 3; i++) { e->box_lin[i] = 0; e->box_ang[i] = 0; }
    e->success_duration_steps = 0; e->success_conditions_met = 0;
    e->current_stage = 1; e->time_in_stage = 0; e->stage_contact_duration = 0; e->just2 = 0; e->just3 = 0;
    for (int i = 0; i < 3; i++) e->q[i] = (u[3 + i] * 2 - 1) * c->hand_translation_range;
    for (int i = 0; i < 3; i++) e->q[3 + i] = (u[6 + i] * 2 - 1) * c->hand_rotation_range;
    for (int i = 0; i < 20; i++) e->q[6 + i] = u[9 + i] * (i == 0 ? c->thumb_rotation_range : c->other_finger_range);
  }
  e->reset_count++;
  e->wgen = DEXSIM_WGEN_NEXT(e->wgen);   /* the warm-start cache does not survive a teleport */
  /* ActionProcessor.reset_targets (action_processor.py:524-568) */
  for (int i = 0; i < NJ; i++) e->targets[i] = e->q[i];
  extract_active_targets(e->q, e->active_prev_targets);
}

static void reset_observer_env(OrcEnv* e) { /* observation_encoder.py:363-383 */
  for (int f = 0; f < 5; f++) { e->contact_duration_steps[f] = 0; e->prev_contact_binary[f] = 0; }
  for (int i = 0; i < NJ; i++) e->prev_dof_pos[i] = 0;
  for (int i = 0; i < 18; i++) e->prev_actions[i] = 0;
}

/* ------------------------------------------------------------------------------------------ orchestration */
#ifdef _OPENMP
#include <omp.h>
#define PAR_FOR _Pragma("omp parallel for schedule(static)")
#else
#define PAR_FOR
#endif

void* orc_create(const DexSimConfig* cfg, const DexHandModel* model) {
  Oracle* o = (Oracle*)calloc(1, sizeof(Oracle));
  o->cfg = *cfg; o->model = *model; o->n = cfg->num_envs;
  o->env = (OrcEnv*)calloc((size_t)o->n, sizeof(OrcEnv));
  o->obs_buf = (float*)calloc((size_t)o->n * cfg->num_obs, sizeof(float));
  o->rc_first_call = 1;
  return o;
}
void orc_destroy(void* h) { Oracle* o = (Oracle*)h; free(o->env); free(o->obs_buf); free(o->reset_samples); free(o); }
void orc_set_threads(void* h, int n) {
  (void)h;
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
void orc_set_reset_samples(void* h, const float* s) {
  Oracle* o = (Oracle*)h;
  free(o->reset_samples); o->reset_samples = NULL;
  if (s) { size_t n = (size_t)o->n * DEXSIM_NRESET_SAMPLES; o->reset_samples = (float*)malloc(n * sizeof(float)); memcpy(o->reset_samples, s, n * sizeof(float)); }
}

void orc_init_state(void* h) {
  Oracle* o = (Oracle*)h;
  const DexSimConfig* c = &o->cfg;
  for (int i = 0; i < o->n; i++) {
    OrcEnv* e = &o->env[i];
    memset(e, 0, sizeof *e);
    e->box_quat[3] = 1; e->box_pos[2] = c->box_z; e->initial_box_pos[2] = c->box_z;
    if (c->box_fixed) for (int k = 0; k < 3; k++) { e->box_pos[k] = c->box_fixed_pos[k]; e->initial_box_pos[k] = c->box_fixed_pos[k]; }
    e->box_mass = c->box_mass > 0 ? c->box_mass : 1; e->box_mu = c->box_friction;
    if (c->dr_enabled) {
      uint32_t x[4];
      philox4x32(x, (uint32_t)i, 0, 0, 0, c->dr_seed, 0xd12au);
      e->box_mass = c->dr_mass_lo + u01(x[0]) * (c->dr_mass_hi - c->dr_mass_lo);
      e->box_mu = c->dr_mu_lo + u01(x[1]) * (c->dr_mu_hi - c->dr_mu_lo);
    }
    e->current_stage = 1;
    publish(o, e);
  }
  o->rc_first_call = 1; o->consecutive_successes = 0; o->any_reset = 0;
  memset(o->stats, 0, sizeof o->stats);
}

void orc_process_actions(void* h, const float* actions, int zero_targets) {
  Oracle* o = (Oracle*)h;
  PAR_FOR
  for (int i = 0; i < o->n; i++) process_actions_env(o, &o->env[i], actions + (size_t)i * o->cfg.num_actions, zero_targets);
}

void orc_physics_step(void* h) {
  Oracle* o = (Oracle*)h;
  PAR_FOR
  for (int i = 0; i < o->n; i++) physics_step_env(o, &o->env[i]);
  o->stats[DEXSIM_STAT_PHYSICS_STEPS] += 1;
  double kc = 0, kh = 0;
  for (int i = 0; i < o->n; i++) {
    kc += o->env[i].ncontact;
    for (int k = 0; k < o->env[i].ncontact; k++) kh += o->env[i].contact[k].type < 2;   /* (joint-limit rows are not contacts) */
  }
  o->stats[DEXSIM_STAT_MEAN_CONTACTS] = (float)(kc / o->n);
  o->stats[DEXSIM_STAT_MEAN_HAND_CONTACTS] = (float)(kh / o->n);
}

/* single sub-step / publish, for teacher-forced parity tests */
void orc_substep(void* h, int last) {
  Oracle* o = (Oracle*)h;
  real hh = o->cfg.dt / (real)o->cfg.substeps;
  PAR_FOR
  for (int i = 0; i < o->n; i++) substep(o, &o->env[i], hh, last);
}
void orc_publish(void* h) {
  Oracle* o = (Oracle*)h;
  PAR_FOR
  for (int i = 0; i < o->n; i++) publish(o, &o->env[i]);
}

static void finalize_stats(Oracle* o) {
  const DexSimConfig* c = &o->cfg;
  double cs[DEXSIM_NUM_SUCC] = {0}, cf[DEXSIM_NUM_FAIL] = {0}, ts = 0, tf = 0, tt = 0, nr = 0;
  for (int i = 0; i < o->n; i++) {
    OrcEnv* e = &o->env[i];
    for (int k = 0; k < DEXSIM_NUM_SUCC; k++) cs[k] += e->crit_success[k];
    for (int k = 0; k < DEXSIM_NUM_FAIL; k++) cf[k] += e->crit_failure[k];
    ts += e->term_success; tf += e->term_failure; tt += e->term_timeout; nr += e->reset_flag;
  }
  for (int k = 0; k < DEXSIM_NUM_SUCC; k++) o->stats[DEXSIM_STAT_SUCC_MEAN + k] = (float)(cs[k] / o->n);
  for (int k = 0; k < DEXSIM_NUM_FAIL; k++) o->stats[DEXSIM_STAT_FAIL_MEAN + k] = (float)(cf[k] / o->n);
  o->stats[DEXSIM_STAT_SUCCESS_RATE] = (float)(ts / o->n);
  o->stats[DEXSIM_STAT_FAILURE_RATE] = (float)(tf / o->n);
  o->stats[DEXSIM_STAT_TIMEOUT_RATE] = (float)(tt / o->n);
  /* update_consecutive_successes (termination_manager.py:323-339) */
  o->consecutive_successes = ts > 0 ? o->consecutive_successes + 1 : 0;
  if (o->consecutive_successes > c->max_consecutive_successes) o->consecutive_successes = c->max_consecutive_successes;
  o->stats[DEXSIM_STAT_CONSECUTIVE_SUCCESSES] = (float)o->consecutive_successes;
  o->stats[DEXSIM_STAT_NUM_RESETS] = (float)nr;
}

/* StepProcessor.process_physics_step (step_processor.py:37-131) */
void orc_post_physics(void* h, int obs_only) {
  Oracle* o = (Oracle*)h;
  int first = o->rc_first_call;
  PAR_FOR
  for (int i = 0; i < o->n; i++) {
    OrcEnv* e = &o->env[i];
    if (!obs_only) refresh_cf5(e);
    compute_observations_env(o, e);
    concat_observations_env(o, e, i);
    if (obs_only) continue;
    e->episode_step += 1;
    evaluate_termination_env(o, e);
    compute_rewards_env(o, e, first);
  }
  if (obs_only) return;
  o->rc_first_call = 0;
  finalize_stats(o);
  int any = 0;
  for (int i = 0; i < o->n; i++) any |= o->env[i].reset_flag;
  o->any_reset = any;
  if (any) { /* reset_idx(nonzero(reset_buf)) incl. the extra physics step for ALL envs */
    PAR_FOR
    for (int i = 0; i < o->n; i++) if (o->env[i].reset_flag) reset_env(o, &o->env[i], i);
    orc_physics_step(o);
    for (int i = 0; i < o->n; i++) if (o->env[i].reset_flag) reset_observer_env(&o->env[i]);
  }
}

void orc_step(void* h, const float* actions) { /* DexHandBase.step (dexhand_base.py:893-942) */
  Oracle* o = (Oracle*)h;
  o->stats[DEXSIM_STAT_PHYSICS_STEPS] = 0;
  orc_process_actions(h, actions, 0);
  orc_physics_step(h);
  /* the contact statistics of a control step are those of its main physics step (not of the conditional extra one) */
  const float kc = o->stats[DEXSIM_STAT_MEAN_CONTACTS], kh = o->stats[DEXSIM_STAT_MEAN_HAND_CONTACTS];
  orc_post_physics(h, 0);
  o->stats[DEXSIM_STAT_MEAN_CONTACTS] = kc; o->stats[DEXSIM_STAT_MEAN_HAND_CONTACTS] = kh;
}

void orc_reset_idx(void* h, const int64_t* ids, int k) { /* DexHandBase.reset_idx (dexhand_base.py:743-803) */
  Oracle* o = (Oracle*)h;
  if (k == 0) return;
  for (int i = 0; i < k; i++) reset_env(o, &o->env[ids[i]], (int)ids[i]);
  orc_physics_step(o);
  for (int i = 0; i < k; i++) reset_observer_env(&o->env[ids[i]]);
}

void orc_reset(void* h) { /* DexHandBase.reset (dexhand_base.py:805-838) */
  Oracle* o = (Oracle*)h;
  o->stats[DEXSIM_STAT_PHYSICS_STEPS] = 0;
  for (int i = 0; i < o->n; i++) reset_env(o, &o->env[i], i);
  orc_physics_step(o);
  for (int i = 0; i < o->n; i++) reset_observer_env(&o->env[i]);
  orc_post_physics(h, 1);
  orc_post_physics(h, 0);
}

/* ------------------------------------------------------------------------------------------ field access
 * Fields are exchanged as SoA [rows][N] float (ints converted), the same shape the HIP arena uses. */
typedef struct FieldDesc { const char* name; int rows; int is_int; size_t off; } FieldDesc;
#define OFF(f) offsetof(OrcEnv, f)
static const FieldDesc FIELDS[] = {
  {"q", NJ, 0, OFF(q)}, {"qd", NJ, 0, OFF(qd)}, {"targets", NJ, 0, OFF(targets)},
  {"box_pos", 3, 0, OFF(box_pos)}, {"box_quat", 4, 0, OFF(box_quat)}, {"box_lin", 3, 0, OFF(box_lin)},
  {"box_ang", 3, 0, OFF(box_ang)}, {"box_mass", 1, 0, OFF(box_mass)}, {"box_mu", 1, 0, OFF(box_mu)},
  {"site_pose", 77, 0, OFF(site_pose)}, {"hand_vel", 6, 0, OFF(hand_vel)}, {"cforce", 51, 0, OFF(cforce)},
  {"cf5", 15, 0, OFF(cf5)},
  {"ncontact", 1, 1, OFF(ncontact)},
  {"wlam", DEXSIM_NWKEY * 3, 0, OFF(wlam)}, {"wtag", DEXSIM_NWKEY, 1, OFF(wtag)}, {"wgen", 1, 1, OFF(wgen)},
  {"active_prev_targets", 18, 0, OFF(active_prev_targets)}, {"active_rule_targets", 18, 0, OFF(active_rule_targets)},
  {"prev_actions", 18, 0, OFF(prev_actions)}, {"actions", 18, 0, OFF(actions)}, {"prev_dof_pos", NJ, 0, OFF(prev_dof_pos)},
  {"contact_duration_steps", 5, 0, OFF(contact_duration_steps)}, {"prev_contact_binary", 5, 0, OFF(prev_contact_binary)},
  {"episode_step", 1, 1, OFF(episode_step)},
  {"success_duration_steps", 1, 1, OFF(success_duration_steps)}, {"success_conditions_met", 1, 1, OFF(success_conditions_met)},
  {"current_stage", 1, 1, OFF(current_stage)}, {"just2", 1, 1, OFF(just2)}, {"just3", 1, 1, OFF(just3)},
  {"time_in_stage", 1, 0, OFF(time_in_stage)}, {"stage_contact_duration", 1, 0, OFF(stage_contact_duration)},
  {"initial_box_pos", 3, 0, OFF(initial_box_pos)},
  {"prev_finger_dof_vel", 20, 0, OFF(prev_finger_dof_vel)}, {"prev_hand_vel", 3, 0, OFF(prev_hand_vel)},
  {"prev_hand_ang_vel", 3, 0, OFF(prev_hand_ang_vel)}, {"prev_contacts", 5, 1, OFF(prev_contacts)},
  {"episode_success", 1, 1, OFF(episode_success)}, {"episode_failure", 1, 1, OFF(episode_failure)},
  {"success_reason", DEXSIM_NUM_SUCC, 1, OFF(success_reason)}, {"failure_reason", DEXSIM_NUM_FAIL, 1, OFF(failure_reason)},
  {"crit_success", DEXSIM_NUM_SUCC, 1, OFF(crit_success)}, {"crit_failure", DEXSIM_NUM_FAIL, 1, OFF(crit_failure)},
  {"term_success", 1, 1, OFF(term_success)}, {"term_failure", 1, 1, OFF(term_failure)}, {"term_timeout", 1, 1, OFF(term_timeout)},
  {"obs_all", DEXSIM_OBS_ALL_DIM, 0, OFF(obs_all)}, {"rew_comp", DEXSIM_NUM_REWROWS, 0, OFF(rew_comp)},
  {"rew", 1, 0, OFF(rew)}, {"reset_flag", 1, 1, OFF(reset_flag)}, {"reset_count", 1, 1, OFF(reset_count)},
};
#define NFIELDS ((int)(sizeof FIELDS / sizeof FIELDS[0]))

static const FieldDesc* find_field(const char* name) {
  for (int i = 0; i < NFIELDS; i++) if (!strcmp(FIELDS[i].name, name)) return &FIELDS[i];
  return NULL;
}
int orc_field_rows(const char* name) { const FieldDesc* f = find_field(name); return f ? f->rows : -1; }
int orc_num_fields(void) { return NFIELDS; }
const char* orc_field_name(int i) { return FIELDS[i].name; }

int orc_get_field(void* h, const char* name, double* out) { /* out: [rows][N] */
  Oracle* o = (Oracle*)h; const FieldDesc* f = find_field(name);
  if (!f) return 1;
  for (int i = 0; i < o->n; i++) {
    const char* base = (const char*)&o->env[i] + f->off;
    for (int r = 0; r < f->rows; r++)
      out[(size_t)r * o->n + i] = f->is_int ? (double)((const int*)base)[r] : (double)((const real*)base)[r];
  }
  return 0;
}
int orc_set_field(void* h, const char* name, const double* in) {
  Oracle* o = (Oracle*)h; const FieldDesc* f = find_field(name);
  if (!f) return 1;
  for (int i = 0; i < o->n; i++) {
    char* base = (char*)&o->env[i] + f->off;
    for (int r = 0; r < f->rows; r++) {
      double v = in[(size_t)r * o->n + i];
      if (f->is_int) ((int*)base)[r] = (int)v; else ((real*)base)[r] = (real)v;
    }
  }
  return 0;
}
/* contact list of env i: rows of [p3, n3, gap, mu, type, cap] */
int orc_get_contacts(void* h, int env, double* out /* KMAX*10 */) {
  Oracle* o = (Oracle*)h; OrcEnv* e = &o->env[env];
  for (int k = 0; k < e->ncontact; k++) {
    OrcContact* c = &e->contact[k];
    double* r = out + 10 * k;
    for (int i = 0; i < 3; i++) { r[i] = c->p[i]; r[3 + i] = c->n[i]; }
    r[6] = c->gap; r[7] = c->mu; r[8] = c->type; r[9] = c->cap;
  }
  return e->ncontact;
}
void orc_get_obs_buf(void* h, float* out) { Oracle* o = (Oracle*)h; memcpy(out, o->obs_buf, (size_t)o->n * o->cfg.num_obs * sizeof(float)); }
void orc_get_stats(void* h, float* out) { Oracle* o = (Oracle*)h; memcpy(out, o->stats, sizeof o->stats); }
int orc_any_reset(void* h) { return ((Oracle*)h)->any_reset; }
int orc_sizeof_real(void) { return (int)sizeof(real); }
void orc_set_rc_first_call(void* h, int v) { ((Oracle*)h)->rc_first_call = v; }

/* stand-alone L2 entry points for the golden-vector tests (state injected through orc_set_field) */
void orc_compute_observations(void* h) {
  Oracle* o = (Oracle*)h;
  for (int i = 0; i < o->n; i++) { compute_observations_env(o, &o->env[i]); concat_observations_env(o, &o->env[i], i); }
}
void orc_l2_step_no_reset(void* h) { /* obs + count + termination + reward, WITHOUT performing the resets */
  Oracle* o = (Oracle*)h;
  int first = o->rc_first_call;
  for (int i = 0; i < o->n; i++) {
    OrcEnv* e = &o->env[i];
    refresh_cf5(e);
    compute_observations_env(o, e);
    concat_observations_env(o, e, i);
    e->episode_step += 1;
    evaluate_termination_env(o, e);
    compute_rewards_env(o, e, first);
  }
  o->rc_first_call = 0;
  finalize_stats(o);
}
void orc_reset_flagged_no_physics(void* h) { /* reset + observer reset of flagged envs, physics step left to the caller */
  Oracle* o = (Oracle*)h;
  for (int i = 0; i < o->n; i++) if (o->env[i].reset_flag) { reset_env(o, &o->env[i], i); reset_observer_env(&o->env[i]); }
}

/* forward kinematics only: joint origins/axes for model sanity tests. out: [26][6] = origin, axis */
void orc_fk(void* h, const double* q, double* out) {
  Oracle* o = (Oracle*)h; real qq[NJ]; FK fk;
  for (int i = 0; i < NJ; i++) qq[i] = (real)q[i];
  fk_compute(&o->model, qq, &fk);
  for (int j = 0; j < NJ; j++) for (int i = 0; i < 3; i++) { out[6 * j + i] = fk.o[j][i]; out[6 * j + 3 + i] = fk.a[j][i]; }
}
/* dense mass matrix and bias at (q, qd) for dynamics sanity tests */
void orc_mass_matrix(void* h, const double* q, const double* qd, double* Mout, double* bias_out) {
  Oracle* o = (Oracle*)h; real qq[NJ], qv[NJ], M[NJ][NJ], b[NJ]; FK fk;
  for (int i = 0; i < NJ; i++) { qq[i] = (real)q[i]; qv[i] = (real)qd[i]; }
  fk_compute(&o->model, qq, &fk);
  crba(&o->model, &fk, M);
  real g[3] = {o->cfg.gravity[0], o->cfg.gravity[1], o->cfg.gravity[2]};
  rnea_bias(&o->model, &fk, qv, g, b);
  for (int i = 0; i < NJ; i++) { bias_out[i] = b[i]; for (int j = 0; j < NJ; j++) Mout[i * NJ + j] = M[i][j]; }
}
