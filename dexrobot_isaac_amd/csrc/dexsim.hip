// dexsim.hip -- libdexsim: C-ABI (include/dexsim.h) over the gfx950 kernels.
//
// Build: hipcc --offload-arch=gfx950 -O3 -fPIC -shared -o libdexsim.so dexsim.hip
// No torch types, no CUDA shims, no dual back-end: this file only targets CDNA4 through HIP.
#include <cmath>
#include <algorithm>
#include "dexsim_device.h"

#include <cstdio>
#include <cstring>
#include <string>

// clang-format off
#include "dexsim_physics.hip.inc"
#include "dexsim_l2.hip.inc"
// clang-format on

static thread_local std::string g_last_error;

static int fail(int code, const char* what) {
  g_last_error = what ? what : "";
  return code;
}
#define HIP_TRY(expr)                                                                 \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess) {                                                           \
      g_last_error = std::string(#expr) + ": " + hipGetErrorString(_e);               \
      return DEXSIM_ERR_HIP;                                                          \
    }                                                                                 \
  } while (0)

struct DexSim {
  DexSimConfig cfg;
  DexHandModel model;
  int device;
  int N;  // real envs
  int NS; // padded to a multiple of 64: arena stride, every lane of every wavefront owns an env
  DevParams* d_params;
  Arena arena;
  ApiPtrs api;
  bool bound;
  hipEvent_t ev0, ev1;
  const float* last_actions;   // device pointer of the last dexsim_step (DEXSIM_STAGE_STEP re-launches the same kernel)
  hipEvent_t tev[128];         // dexsim_step_timing: ring of 64 (start, stop) pairs
  int timing;                  // 0 = off, else 1 + number of steps recorded since it was switched on
};

static int padded(int n) { return (n + 63) / 64 * 64; }

// A new control step begins: new stamp for the device-side reset gate and the parity of the contact statistics (CNT_ANY_RESET,
// CNT_CONTACTS in dexsim_device.h).  Never 0 (the counters block is zero-initialised).
static void next_stamp(DexSim* h) { h->api.stamp = h->api.stamp >= 0x3ffffffe ? 1 : h->api.stamp + 1; }

// Every entry point runs on the handle's device whatever the caller's current device is, and restores the caller's.
struct DeviceGuard {
  int prev = -1; bool switched = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};

static const char* const kObsKeyNames[] = {
#define X(name, dim) #name,
    DEXSIM_OBS_KEYS(X)
#undef X
};
static const int kObsKeyDims[] = {
#define X(name, dim) dim,
    DEXSIM_OBS_KEYS(X)
#undef X
};
static const char* const kRewardNames[] = {
#define X(name) #name,
    DEXSIM_REWARD_TERMS(X)
#undef X
};

extern "C" {

int dexsim_struct_sizes(size_t out[4]) {
  out[0] = sizeof(DexHandModel); out[1] = sizeof(DexSimConfig); out[2] = sizeof(DexSimField); out[3] = sizeof(DexSimBuffers);
  return DEXSIM_OK;
}

int dexsim_arena_layout(const DexSimConfig* cfg, DexSimField* fields, int max_fields, int* n_fields, size_t* arena_words) {
  if (!cfg || !n_fields || !arena_words || cfg->num_envs <= 0) return fail(DEXSIM_ERR_ARG, "dexsim_arena_layout: bad argument");
  const size_t NS = (size_t)padded(cfg->num_envs);
  size_t off = 0;
  int n = 0;
#define X(ty, nm, nrows)                                                      \
  {                                                                           \
    if (fields) {                                                             \
      if (n >= max_fields) return fail(DEXSIM_ERR_LAYOUT, "field table too small"); \
      std::snprintf(fields[n].name, sizeof fields[n].name, "%s", #nm);        \
      fields[n].rows = (nrows);                                               \
      fields[n].is_int = std::string(#ty) == "int"; \
      fields[n].offset = off;                                                 \
    }                                                                         \
    off += (size_t)(nrows) * NS;                                              \
    n++;                                                                      \
  }
  DEXSIM_FIELDS(X)
#undef X
  *n_fields = n;
  *arena_words = off;
  return DEXSIM_OK;
}

int dexsim_obs_key_info(int i, const char** name, int* offset, int* dim) {
  if (i < 0 || i >= DEXSIM_NUM_OBS_KEYS) return fail(DEXSIM_ERR_ARG, "obs key index out of range");
  int off = 0;
  for (int k = 0; k < i; k++) off += kObsKeyDims[k];
  if (name) *name = kObsKeyNames[i];
  if (offset) *offset = off;
  if (dim) *dim = kObsKeyDims[i];
  return DEXSIM_OK;
}

int dexsim_reward_term_name(int i, const char** name) {
  if (i < 0 || i >= DEXSIM_NUM_REWARD_TERMS || !name) return fail(DEXSIM_ERR_ARG, "reward term index out of range");
  *name = kRewardNames[i];
  return DEXSIM_OK;
}

int dexsim_body_name(int i, const char** name) {
  static char names[DEXSIM_NUM_HAND_BODIES][24];
  static bool init = false;
  if (!init) {
    const char* base[7] = {"hand_mount", "ARTx_link", "ARTy_link", "ARTz_link", "ARRx_link", "ARRy_link", "right_hand_base"};
    for (int b = 0; b < 7; b++) std::snprintf(names[b], 24, "%s", base[b]);
    const char* suf[6] = {"1", "2", "3", "4", "pad", "tip"};
    for (int f = 0; f < 5; f++)
      for (int l = 0; l < 6; l++) std::snprintf(names[7 + 6 * f + l], 24, "r_f_link%d_%s", f + 1, suf[l]);
    init = true;
  }
  if (i < 0 || i >= DEXSIM_NUM_HAND_BODIES || !name) return fail(DEXSIM_ERR_ARG, "body index out of range");
  *name = names[i];
  return DEXSIM_OK;
}

static int validate_model(const DexHandModel& m) {
  for (int j = 0; j < DEXSIM_NJ; j++)
    if (m.jtype[j] != (j < 3 ? 0 : 1)) return fail(DEXSIM_ERR_ARG, "model: kernels assume joints 0-2 prismatic, 3-25 revolute");
  for (int j = 0; j < 5; j++)
    if (m.mass[j] != 0.f) return fail(DEXSIM_ERR_ARG, "model: base-chain links 0-4 must be massless (palm rides on joint 5)");
  for (int j = 5; j < DEXSIM_NJ; j++)
    if (!(m.mass[j] > 0.f)) return fail(DEXSIM_ERR_ARG, "model: palm and finger links need positive mass");
  for (int c = 0; c < DEXSIM_NCAP; c++) {
    const int want = c < 3 ? 5 : 6 + 4 * ((c - 3) / 3) + 1 + (c - 3) % 3;
    if (m.cap_parent[c] != want) return fail(DEXSIM_ERR_ARG, "model: capsule c must ride on palm (c<3) or finger link (c-3)/3,(c-3)%3+1");
    if (m.cap_fslot[c] < 0 || m.cap_fslot[c] >= DEXSIM_FSLOT_BOX) return fail(DEXSIM_ERR_ARG, "model: bad capsule force slot");
  }
  if (m.site_parent[0] != 5) return fail(DEXSIM_ERR_ARG, "model: site 0 (right_hand_base) must ride on joint 5");
  for (int f = 0; f < 5; f++)
    if (m.site_parent[1 + f] != 9 + 4 * f || m.site_parent[6 + f] != 9 + 4 * f)
      return fail(DEXSIM_ERR_ARG, "model: tip/pad sites must ride on the distal joint of their finger");
  for (int j = 0; j < DEXSIM_NJ; j++)
    if (!(m.hi[j] > m.lo[j])) return fail(DEXSIM_ERR_ARG, "model: joint range must be non-empty");
  return DEXSIM_OK;
}

int dexsim_create(const DexSimConfig* cfg, const DexHandModel* model, int device, dexsim_t* out) {
  if (!cfg || !model || !out) return fail(DEXSIM_ERR_ARG, "dexsim_create: null argument");
  if (cfg->num_envs <= 0 || cfg->substeps <= 0 || !(cfg->dt > 0.f)) return fail(DEXSIM_ERR_ARG, "dexsim_create: bad num_envs/substeps/dt");
  if (cfg->num_obs <= 0 || cfg->n_obs_seg <= 0 || cfg->n_obs_seg > DEXSIM_MAX_OBS_SEG) return fail(DEXSIM_ERR_ARG, "dexsim_create: bad observation table");
  if (cfg->num_actions != 6 * (cfg->policy_controls_base != 0) + 12 * (cfg->policy_controls_fingers != 0) || cfg->num_actions == 0)
    return fail(DEXSIM_ERR_ARG, "dexsim_create: num_actions inconsistent with policy_controls_*");
  if (cfg->has_box && !(cfg->box_size > 0.f && (cfg->box_mass > 0.f || cfg->box_fixed))) return fail(DEXSIM_ERR_ARG, "dexsim_create: bad box");
  if (cfg->box_fixed && !cfg->has_box) return fail(DEXSIM_ERR_ARG, "dexsim_create: box_fixed needs has_box");
  {
    int tot = 0;
    for (int s = 0; s < cfg->n_obs_seg; s++) {
      if (cfg->obs_seg_off[s] < 0 || cfg->obs_seg_len[s] <= 0 || cfg->obs_seg_off[s] + cfg->obs_seg_len[s] > DEXSIM_OBS_ALL_DIM)
        return fail(DEXSIM_ERR_ARG, "dexsim_create: observation segment out of range");
      tot += cfg->obs_seg_len[s];
    }
    if (tot != cfg->num_obs) return fail(DEXSIM_ERR_ARG, "dexsim_create: num_obs != sum of segments");
  }
  int rc = validate_model(*model);
  if (rc) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(DEXSIM_ERR_NO_DEVICE, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(DEXSIM_ERR_NO_DEVICE, "device index out of range");
  DeviceGuard guard(device);
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur != device) return fail(DEXSIM_ERR_HIP, "dexsim_create: cannot select the device");
  DexSim* h = new DexSim();
  struct Cleanup {   // a failure below must not leak the handle, the parameter block or the events
    DexSim* h; bool armed = true;
    ~Cleanup() {
      if (!armed) return;
      if (h->d_params) (void)hipFree(h->d_params);
      if (h->ev0) (void)hipEventDestroy(h->ev0);
      if (h->ev1) (void)hipEventDestroy(h->ev1);
      delete h;
    }
  } cleanup{h};
  h->cfg = *cfg; h->model = *model; h->device = device;
  h->api.stamp = 1;
  h->N = cfg->num_envs; h->NS = padded(cfg->num_envs);
  h->bound = false;
  DevParams hp;
  hp.cfg = *cfg; hp.model = *model;
  hp.h = cfg->dt / (float)cfg->substeps;
  hp.box_inv_I_k = cfg->has_box ? 6.f / (cfg->box_size * cfg->box_size) : 0.f;
  hp.obs_div_magic = (unsigned)((0x100000000ull + (unsigned long long)cfg->num_obs - 1) / (unsigned long long)cfg->num_obs);
  {   // base-chain constants (double precision on the host): see base_chain()
    struct Qd { double x, y, z, w; };
    auto qmul = [](Qd a, Qd b) {
      return Qd{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
                a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
    };
    auto qrot = [&](Qd q, const double* v, double* out) {   // R(q) v
      const Qd p{v[0], v[1], v[2], 0.0}, c{-q.x, -q.y, -q.z, q.w};
      const Qd r = qmul(qmul(q, p), c);
      out[0] = r.x; out[1] = r.y; out[2] = r.z;
    };
    Qd qc{model->spawn_quat[0], model->spawn_quat[1], model->spawn_quat[2], model->spawn_quat[3]};
    double oc[3] = {model->spawn_pos[0], model->spawn_pos[1], model->spawn_pos[2]};
    for (int j = 0; j < 4; j++) {
      const double poff[3] = {model->jpoff[j][0], model->jpoff[j][1], model->jpoff[j][2]};
      const double ax[3] = {model->jaxis[j][0], model->jaxis[j][1], model->jaxis[j][2]};
      double d[3], aw[3];
      qrot(qc, poff, d);
      for (int i = 0; i < 3; i++) oc[i] += d[i];
      const Qd qz = qmul(qc, Qd{model->jqoff[j][0], model->jqoff[j][1], model->jqoff[j][2], model->jqoff[j][3]});
      qrot(qz, ax, aw);
      for (int i = 0; i < 3; i++) { hp.base_A[j][i] = (float)aw[i]; hp.base_Oc[j][i] = (float)oc[i]; }
      qc = qz;   // prismatic joints keep the orientation; for joint 3 this is the frame at q3 = 0
    }
    hp.base_Q3z[0] = (float)qc.x; hp.base_Q3z[1] = (float)qc.y; hp.base_Q3z[2] = (float)qc.z; hp.base_Q3z[3] = (float)qc.w;
  }
  {   // hand-level broadphase radius: chain of joint offsets from the palm to the capsule's joint + capsule extent + radius
    auto len3 = [](const float* v) { return std::sqrt((double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2]); };
    double reach = 0.0;
    for (int c = 0; c < DEXSIM_NCAP; c++) {
      const int j = model->cap_parent[c];
      double d = std::max(len3(model->cap_p0[c]), len3(model->cap_p1[c])) + model->cap_r[c];
      if (j >= 6) for (int l = 6 + 4 * ((j - 6) / 4); l <= j; l++) d += len3(model->jpoff[l]);
      reach = std::max(reach, d);
    }
    hp.hand_reach = (float)(reach * 1.001 + 1e-4);
  }
  {
    int k = 0;
    for (int s = 0; s < cfg->n_obs_seg; s++)
      for (int i = 0; i < cfg->obs_seg_len[s]; i++) hp.obs_col_row[k++] = cfg->obs_seg_off[s] + i;
    for (; k < DEXSIM_OBS_ALL_DIM; k++) hp.obs_col_row[k] = 0;
  }
  for (int j = 0; j < DEXSIM_NJ; j++) {
    JC& c = hp.jc[j];
    std::memset(&c, 0, sizeof c);
    for (int i = 0; i < 4; i++) c.qoff[i] = model->jqoff[j][i];
    for (int i = 0; i < 3; i++) { c.poff[i] = model->jpoff[j][i]; c.axis[i] = model->jaxis[j][i]; c.com[i] = model->com[j][i]; }
    for (int i = 0; i < 6; i++) c.inertia[i] = model->inertia[j][i];
    c.mass = model->mass[j]; c.kp = model->kp[j]; c.kd = model->kd[j]; c.armature = model->armature[j];
    c.lo = model->lo[j]; c.hi = model->hi[j];
  }
  hp.inertia_diag = 1;
  for (int j = 6; j < DEXSIM_NJ; j++)
    for (int i = 3; i < 6; i++) if (model->inertia[j][i] != 0.f) hp.inertia_diag = 0;
  HIP_TRY(hipMalloc(&h->d_params, sizeof(DevParams)));
  HIP_TRY(hipMemcpy(h->d_params, &hp, sizeof(DevParams), hipMemcpyHostToDevice));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_solve<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_solve<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_substep<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_substep<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_substep<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_substep<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_post), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_physics4<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_physics4<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(hipEventCreate(&h->ev0));
  HIP_TRY(hipEventCreate(&h->ev1));
  cleanup.armed = false;
  *out = h;
  return DEXSIM_OK;
}

int dexsim_destroy(dexsim_t h) {
  if (!h) return DEXSIM_OK;
  DeviceGuard guard(h->device);
  (void)hipFree(h->d_params);
  (void)hipEventDestroy(h->ev0);
  (void)hipEventDestroy(h->ev1);
  if (h->tev[0]) for (int i = 0; i < 128; i++) (void)hipEventDestroy(h->tev[i]);
  delete h;
  return DEXSIM_OK;
}

int dexsim_bind(dexsim_t h, const DexSimBuffers* b) {
  if (!h || !b) return fail(DEXSIM_ERR_ARG, "dexsim_bind: null argument");
  DeviceGuard guard(h->device);
  if (!b->arena || !b->stats || !b->counters || !b->obs_buf || !b->rew_buf || !b->reset_buf || !b->episode_step_count ||
      !b->episode_length || !b->dof_state || !b->root_state)
    return fail(DEXSIM_ERR_ARG, "dexsim_bind: arena, stats, counters, obs_buf, rew_buf, reset_buf, episode_step_count, "
                                "episode_length, dof_state and root_state are required");
  size_t off = 0;
  const size_t NS = (size_t)h->NS;
  char* base = (char*)b->arena;
#define X(ty, nm, nrows)                           \
  h->arena.nm = (ty*)(base + off * 4);           \
  off += (size_t)(nrows) * NS;
  DEXSIM_FIELDS(X)
#undef X
  h->api.obs_buf = b->obs_buf; h->api.rew_buf = b->rew_buf; h->api.reset_buf = b->reset_buf;
  h->api.episode_step_count = b->episode_step_count; h->api.episode_length = b->episode_length;
  h->api.dof_state = b->dof_state; h->api.root_state = b->root_state;
  h->api.rigid_body_states = b->rigid_body_states; h->api.contact_forces_all = b->contact_forces_all;
  h->api.full_dof_targets = b->full_dof_targets; h->api.reset_samples = b->reset_samples;
  h->api.masks = b->masks; h->api.raw_targets = b->raw_targets;
  h->api.stats = b->stats; h->api.counters = b->counters;
  HIP_TRY(hipMemcpy(&h->d_params->arena, &h->arena, sizeof(Arena), hipMemcpyHostToDevice));
  h->bound = true;
  return DEXSIM_OK;
}

#define NEED_BOUND(h)                                                        \
  if (!(h)) return fail(DEXSIM_ERR_ARG, "null handle");                      \
  if (!(h)->bound) return fail(DEXSIM_ERR_NOT_BOUND, "dexsim_bind has not been called"); \
  DeviceGuard _device_guard((h)->device)
#define GRID(h) dim3((h)->NS / 64), dim3(64), 0, (hipStream_t)stream
#define LAUNCH_CHECK() HIP_TRY(hipGetLastError())

static int launch_publish(dexsim_t h, int gate, int full, void* stream) {
  if (gate) k_publish<true><<<dim3(h->NS / 64), dim3(384), 0, (hipStream_t)stream>>>(h->arena, h->api, h->d_params, h->api.counters, full, h->NS, h->N);
  else k_publish<false><<<dim3(h->NS / 64), dim3(384), 0, (hipStream_t)stream>>>(h->arena, h->api, h->d_params, h->api.counters, full, h->NS, h->N);
  LAUNCH_CHECK();
  return DEXSIM_OK;
}

// LDS budget of the contact-solve kernel: stage the rows of as many contacts as fit next to u_f and the impulses.
// At <= 1 wavefront per CU (num_envs <= 64 * #CUs) the whole 160 KiB is this wave's to use.
static int solve_kstage(const DexSim* h) { return h->NS <= 64 * 256 ? 4 : 2; }   // 4: (60 + 7 KMAX + 4 x 84) words x 256 B = 141 KiB
static size_t solve_lds_bytes(int kstage) { return (size_t)SOLVE_LDS_WORDS(kstage) * 64 * sizeof(float); }

static int launch_solve(dexsim_t h, int gate, int last, void* stream) {
  const int ks = solve_kstage(h);
  const size_t lds = solve_lds_bytes(ks);
  if (gate) k_solve<true><<<dim3(h->NS / 64), dim3(64), lds, (hipStream_t)stream>>>(h->arena, h->d_params, h->api.counters, h->api.stamp, last, ks, h->NS, h->N);
  else k_solve<false><<<dim3(h->NS / 64), dim3(64), lds, (hipStream_t)stream>>>(h->arena, h->d_params, h->api.counters, h->api.stamp, last, ks, h->NS, h->N);
  LAUNCH_CHECK();
  return DEXSIM_OK;
}
// one fused sub-step; `last` adds the contact-force accumulation and the publication
static int launch_substep(dexsim_t h, int gate, int last, void* stream) {
  const size_t lds = (size_t)FS_WORDS * 64 * sizeof(float);
  const dim3 grid(h->NS / 64), block(448);
  hipStream_t st = (hipStream_t)stream;
  if (gate && last) k_substep<true, true><<<grid, block, lds, st>>>(h->arena, h->api, h->d_params, h->api.counters, h->NS, h->N);
  else if (gate) k_substep<true, false><<<grid, block, lds, st>>>(h->arena, h->api, h->d_params, h->api.counters, h->NS, h->N);
  else if (last) k_substep<false, true><<<grid, block, lds, st>>>(h->arena, h->api, h->d_params, h->api.counters, h->NS, h->N);
  else k_substep<false, false><<<grid, block, lds, st>>>(h->arena, h->api, h->d_params, h->api.counters, h->NS, h->N);
  LAUNCH_CHECK();
  return DEXSIM_OK;
}
static int launch_dynamics(dexsim_t h, int gate, void* stream) {
  if (gate) k_dynamics<true><<<dim3(h->NS / 64), dim3(384), 0, (hipStream_t)stream>>>(h->arena, h->d_params, h->api.counters, h->api.stamp, h->NS);
  else k_dynamics<false><<<dim3(h->NS / 64), dim3(384), 0, (hipStream_t)stream>>>(h->arena, h->d_params, h->api.counters, h->api.stamp, h->NS);
  LAUNCH_CHECK();
  return DEXSIM_OK;
}

int dexsim_init_state(dexsim_t h, void* stream) {
  NEED_BOUND(h);
  size_t words = 0; int nf = 0;
  dexsim_arena_layout(&h->cfg, nullptr, 0, &nf, &words);
  HIP_TRY(hipMemsetAsync(h->arena.q, 0, words * 4, (hipStream_t)stream));
  HIP_TRY(hipMemsetAsync(h->api.episode_step_count, 0, sizeof(int64_t) * h->N, (hipStream_t)stream));
  HIP_TRY(hipMemsetAsync(h->api.episode_length, 0, sizeof(int64_t) * h->N, (hipStream_t)stream));
  HIP_TRY(hipMemsetAsync(h->api.reset_buf, 0, h->N, (hipStream_t)stream));
  HIP_TRY(hipMemsetAsync(h->api.rew_buf, 0, sizeof(float) * h->N, (hipStream_t)stream));
  h->api.stamp = 1;
  k_init<<<GRID(h)>>>(h->arena, h->api, h->d_params, h->NS, h->N);
  LAUNCH_CHECK();
  return launch_publish(h, 0, 0, stream);
}

int dexsim_process_actions(dexsim_t h, const float* actions, int zero_targets, void* stream) {
  NEED_BOUND(h);
  if (!actions) return fail(DEXSIM_ERR_ARG, "Actions cannot be None");   // action_processor.py:296-297
  next_stamp(h);   // the action stage opens a control step
  ApiPtrs api = h->api;
  if (api.actions_copy == actions) api.actions_copy = nullptr;   // the caller passed the bound copy itself: nothing to copy
  k_actions<<<GRID(h)>>>(h->arena, api, h->d_params, actions, zero_targets, h->NS, h->N);
  LAUNCH_CHECK();
  return DEXSIM_OK;
}

// tail != 0 (gated launch of the step path only): the launch also applies phase 1 of the in-step reset and finalises
// the step statistics, so that a control step is k_actions, k_physics4, k_post, k_physics4<gated> and nothing else
static int physics_step(dexsim_t h, int gate_on_reset, int tail, void* stream, const float* actions = nullptr) {
  if (h->cfg.substeps % 4 == 0) {
    // the reference's setting (4): the whole sim.dt in one launch.  Multiples of 4 (round 3: cfg/physics/accurate.yaml = 32): one
    // launch per four sub-steps -- the action block rides on the first launch, the post-physics block / the reset tail on the
    // last; launches before the last carry tail bit 2 ("not final": their fourth body does not add to the contact statistics)
    const size_t lds = (size_t)FS_WORDS * 64 * sizeof(float);
    const dim3 grid(h->NS / 64), block(448);
    const int n4 = h->cfg.substeps / 4;
    for (int i = 0; i < n4; i++) {
      const int t = i == n4 - 1 ? tail : 4;
      const float* act = i == 0 ? actions : nullptr;
      if (gate_on_reset) k_physics4<true><<<grid, block, lds, (hipStream_t)stream>>>(h->arena, h->api, h->d_params, h->api.counters, nullptr, t, h->NS, h->N);
      else {
        ApiPtrs api = h->api;
        if (api.actions_copy == act) api.actions_copy = nullptr;   // `act` is __restrict__: never alias it with the copy sink
        k_physics4<false><<<grid, block, lds, (hipStream_t)stream>>>(h->arena, api, h->d_params, h->api.counters, act, t, h->NS, h->N);
      }
      LAUNCH_CHECK();
    }
    return DEXSIM_OK;
  }
  // other sub-step counts: `substeps` fused launches (dynamics + contact solve + integrate); the last one also publishes
  for (int s = 0; s < h->cfg.substeps; s++) {
    int rc = launch_substep(h, gate_on_reset, s == h->cfg.substeps - 1, stream);
    if (rc) return rc;
  }
  return DEXSIM_OK;
}

int dexsim_physics_step(dexsim_t h, int gate_on_reset, void* stream) {
  NEED_BOUND(h);
  return physics_step(h, gate_on_reset, 0, stream);
}

int dexsim_post_physics(dexsim_t h, int obs_only, void* stream) {
  NEED_BOUND(h);
  const int fused = h->cfg.substeps % 4 == 0 && !obs_only;
  k_post<<<dim3(h->NS / 64), dim3(512), POST_LDS_BYTES, (hipStream_t)stream>>>(h->arena, h->api, h->d_params, h->api.counters, obs_only, fused, h->NS, h->N);
  LAUNCH_CHECK();
  if (obs_only) return DEXSIM_OK;
  // reset_idx(nonzero(reset_buf)) incl. the extra physics step for ALL envs (step_processor.py:109-111,
  // reset_manager.py:180), gated on the device-side flag instead of torch.any() on the host
  if (fused) return physics_step(h, 1, 1, stream);   // phase 0 ran inside k_post, phase 1 + statistics run in the gated launch
  k_reset<<<GRID(h)>>>(h->arena, h->api, h->d_params, h->api.counters, nullptr, 0, 0, 0, h->NS, h->N);
  LAUNCH_CHECK();
  int rc = physics_step(h, 1, 0, stream);
  if (rc) return rc;
  // phase 1 of the masked reset; its first thread also finalises the step statistics (k_finalize folded in)
  k_reset<<<GRID(h)>>>(h->arena, h->api, h->d_params, h->api.counters, nullptr, 0, 0, 3, h->NS, h->N);
  LAUNCH_CHECK();
  return DEXSIM_OK;
}

int dexsim_step(dexsim_t h, const float* actions, void* stream) {
  NEED_BOUND(h);
  if (h->cfg.substeps % 4 == 0) {
    // actions + physics + post-physics (+ phase 0 of the in-step reset) in one launch, then the device-gated extra
    // physics step with phase 1 of the reset and the step statistics: a control step is 2 launches
    if (!actions) return fail(DEXSIM_ERR_ARG, "Actions cannot be None");   // action_processor.py:296-297
    next_stamp(h);
    h->last_actions = actions;
    const int slot = h->timing ? ((h->timing - 1) & 63) : -1;
    if (slot >= 0) HIP_TRY(hipEventRecord(h->tev[2 * slot], (hipStream_t)stream));
    int rc = physics_step(h, 0, 2, stream, actions);
    if (rc) return rc;
    if (slot >= 0) { HIP_TRY(hipEventRecord(h->tev[2 * slot + 1], (hipStream_t)stream)); h->timing++; }
    return physics_step(h, 1, 1, stream);
  }
  int rc = dexsim_process_actions(h, actions, 0, stream);   // also clears the per-step device flags
  if (rc) return rc;
  rc = dexsim_physics_step(h, 0, stream);
  if (rc) return rc;
  return dexsim_post_physics(h, 0, stream);
}

int dexsim_reset_idx(dexsim_t h, const int64_t* env_ids, int k, void* stream) {
  NEED_BOUND(h);
  if (k == 0) return DEXSIM_OK;   // dexhand_base.py:746-747
  if (k < 0 || (!env_ids && k != h->N)) return fail(DEXSIM_ERR_ARG, "dexsim_reset_idx: bad env_ids");
  const int mode = env_ids ? 1 : 2;
  dim3 grid((k + 63) / 64);
  k_reset<<<grid, 64, 0, (hipStream_t)stream>>>(h->arena, h->api, h->d_params, h->api.counters, env_ids, k, mode, 0, h->NS, h->N);
  LAUNCH_CHECK();
  int rc = dexsim_physics_step(h, 0, stream);
  if (rc) return rc;
  k_reset<<<grid, 64, 0, (hipStream_t)stream>>>(h->arena, h->api, h->d_params, h->api.counters, env_ids, k, mode, 1, h->NS, h->N);
  LAUNCH_CHECK();
  return DEXSIM_OK;
}

int dexsim_reset(dexsim_t h, void* stream) {
  NEED_BOUND(h);
  next_stamp(h);
  k_begin_step<<<1, 64, 0, (hipStream_t)stream>>>(h->api.counters, h->api.stamp);
  LAUNCH_CHECK();
  int rc = dexsim_reset_idx(h, nullptr, h->N, stream);
  if (rc) return rc;
  rc = dexsim_post_physics(h, 1, stream);
  if (rc) return rc;
  return dexsim_post_physics(h, 0, stream);
}

int dexsim_refresh_body_states(dexsim_t h, void* stream) {
  NEED_BOUND(h);
  if (!h->api.rigid_body_states || !h->api.contact_forces_all) return fail(DEXSIM_ERR_NOT_BOUND, "rigid_body_states / contact_forces_all not bound");
  return launch_publish(h, 0, 1, stream);
}

static int ingest(dexsim_t h, const int64_t* ids, int k, int what, void* stream) {
  NEED_BOUND(h);
  if (k == 0) return DEXSIM_OK;
  if (!ids || k < 0) return fail(DEXSIM_ERR_ARG, "indexed set: bad env_ids");
  k_ingest<<<dim3((k + 63) / 64), 64, 0, (hipStream_t)stream>>>(h->arena, h->api, h->d_params, ids, k, what, h->NS, h->N);
  LAUNCH_CHECK();
  return launch_publish(h, 0, 0, stream);
}
int dexsim_set_dof_state_indexed(dexsim_t h, const int64_t* env_ids, int k, void* stream) { return ingest(h, env_ids, k, 0, stream); }
int dexsim_set_root_state_indexed(dexsim_t h, const int64_t* env_ids, int k, void* stream) { return ingest(h, env_ids, k, 1, stream); }

static int launch_stage(dexsim_t h, int stage, void* stream) {
  if (h->cfg.joint_limit_rows && (stage == DEXSIM_STAGE_DYNAMICS || stage == DEXSIM_STAGE_SOLVE))
    return fail(DEXSIM_ERR_ARG, "the un-fused k_dynamics / k_solve test kernels do not build joint-limit rows (joint_limit_rows): use the fused path");
  switch (stage) {
    case DEXSIM_STAGE_DYNAMICS: return launch_dynamics(h, 0, stream);
    case DEXSIM_STAGE_SOLVE: return launch_solve(h, 0, 1, stream);
    case DEXSIM_STAGE_PUBLISH: return launch_publish(h, 0, 0, stream);
    case DEXSIM_STAGE_SUBSTEP: return launch_substep(h, 0, 1, stream);
    case DEXSIM_STAGE_PHYSICS: return physics_step(h, 0, 0, stream);
    case DEXSIM_STAGE_STEP:   // the production launch of dexsim_step: actions + 4 sub-steps + post-physics, same arguments
      if (h->cfg.substeps != 4 || !h->last_actions) return fail(DEXSIM_ERR_ARG, "DEXSIM_STAGE_STEP needs substeps == 4 and a previous dexsim_step");
      return physics_step(h, 0, 2, stream, h->last_actions);
    case DEXSIM_STAGE_POST: k_post<<<dim3(h->NS / 64), dim3(512), POST_LDS_BYTES, (hipStream_t)stream>>>(h->arena, h->api, h->d_params, h->api.counters, 0, 0, h->NS, h->N); break;
    case DEXSIM_STAGE_POST + 100: k_post<<<dim3(h->NS / 64), dim3(512), POST_LDS_BYTES, (hipStream_t)stream>>>(h->arena, h->api, h->d_params, h->api.counters, 1, 0, h->NS, h->N); break;
    case DEXSIM_STAGE_RESET:   // both phases for the flagged envs, without the device-side gate and without physics
      k_reset<<<GRID(h)>>>(h->arena, h->api, h->d_params, h->api.counters, nullptr, 0, 3, 0, h->NS, h->N);
      k_reset<<<GRID(h)>>>(h->arena, h->api, h->d_params, h->api.counters, nullptr, 0, 3, 1, h->NS, h->N);
      break;
    case DEXSIM_STAGE_FINALIZE:   // closes a staged control step and opens the next one
      k_finalize<<<1, 64, 0, (hipStream_t)stream>>>(h->api, h->d_params, h->N);
      next_stamp(h);
      k_begin_step<<<1, 64, 0, (hipStream_t)stream>>>(h->api.counters, h->api.stamp);
      break;
    default: return fail(DEXSIM_ERR_ARG, "unknown stage");
  }
  LAUNCH_CHECK();
  return DEXSIM_OK;
}

int dexsim_run_stage(dexsim_t h, int stage, void* stream) {
  NEED_BOUND(h);
  return launch_stage(h, stage, stream);
}

int dexsim_time_stage(dexsim_t h, int stage, int launches, void* stream, float* mean_us) {
  NEED_BOUND(h);
  if (launches <= 0 || !mean_us) return fail(DEXSIM_ERR_ARG, "dexsim_time_stage: bad argument");
  double total = 0;
  for (int i = 0; i < launches; i++) {
    // keep the state physical: every timed solve is preceded by its (untimed) dynamics launch
    if (stage == DEXSIM_STAGE_SOLVE) { int rc = launch_stage(h, DEXSIM_STAGE_DYNAMICS, stream); if (rc) return rc; }
    HIP_TRY(hipEventRecord(h->ev0, (hipStream_t)stream));
    int rc = launch_stage(h, stage, stream);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(h->ev1, (hipStream_t)stream));
    HIP_TRY(hipEventSynchronize(h->ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    total += ms;
  }
  *mean_us = (float)(total * 1000.0 / launches);
  return DEXSIM_OK;
}

int dexsim_begin_step(dexsim_t h, void* stream) {
  NEED_BOUND(h);
  next_stamp(h);
  k_begin_step<<<1, 64, 0, (hipStream_t)stream>>>(h->api.counters, h->api.stamp);
  LAUNCH_CHECK();
  return DEXSIM_OK;
}

int dexsim_set_step_sink(dexsim_t h, float* obs, float* rew, uint8_t* done) {
  if (!h) return fail(DEXSIM_ERR_ARG, "null handle");
  h->api.sink_obs = obs; h->api.sink_rew = rew; h->api.sink_done = done;   // kernel arguments: effective from the next launch
  return DEXSIM_OK;
}

int dexsim_set_obs_dict_mode(dexsim_t h, int mode) {
  if (!h) return fail(DEXSIM_ERR_ARG, "null handle");
  if (mode != 0 && mode != 1) return fail(DEXSIM_ERR_ARG, "dexsim_set_obs_dict_mode: mode must be 0 (all rows) or 1 (policy keys only)");
  h->api.skip_obs_all = mode;   // kernel argument: effective from the next launch
  return DEXSIM_OK;
}

int dexsim_set_phase_probe(dexsim_t h, uint32_t* buf) {
  if (!h) return fail(DEXSIM_ERR_ARG, "null handle");
  h->api.probe = buf;   // kernel argument: effective from the next launch
  return DEXSIM_OK;
}

int dexsim_set_stats_sink(dexsim_t h, float* dst) {
  if (!h) return fail(DEXSIM_ERR_ARG, "null handle");
  h->api.sink_stats = dst;   // kernel argument: effective from the next launch
  return DEXSIM_OK;
}

int dexsim_set_action_copy(dexsim_t h, float* dst) {
  if (!h) return fail(DEXSIM_ERR_ARG, "null handle");
  h->api.actions_copy = dst;
  return DEXSIM_OK;
}

int dexsim_step_timing(dexsim_t h, int enable, float* mean_us, int* n) {
  NEED_BOUND(h);
  if (enable) {
    if (!h->tev[0]) for (int i = 0; i < 128; i++) HIP_TRY(hipEventCreate(&h->tev[i]));
    h->timing = 1;
    return DEXSIM_OK;
  }
  const int rec = h->timing ? h->timing - 1 : 0;
  h->timing = 0;
  const int cnt = rec < 64 ? rec : 64;
  double tot = 0.0;
  for (int i = 0; i < cnt; i++) {
    HIP_TRY(hipEventSynchronize(h->tev[2 * i + 1]));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, h->tev[2 * i], h->tev[2 * i + 1]));
    tot += ms;
  }
  if (mean_us) *mean_us = cnt ? (float)(tot * 1000.0 / cnt) : 0.f;
  if (n) *n = cnt;
  return DEXSIM_OK;
}

const char* dexsim_error_string(int code) {
  switch (code) {
    case DEXSIM_OK: return "ok";
    case DEXSIM_ERR_ARG: return "invalid argument";
    case DEXSIM_ERR_NOT_BOUND: return "buffers not bound";
    case DEXSIM_ERR_HIP: return "HIP runtime error";
    case DEXSIM_ERR_NO_DEVICE: return "no usable HIP device";
    case DEXSIM_ERR_LAYOUT: return "arena layout error";
    default: return "unknown error";
  }
}

const char* dexsim_last_error(void) { return g_last_error.c_str(); }

} // extern "C"
