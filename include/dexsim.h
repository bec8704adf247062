/*
 * dexsim.h -- C ABI of libdexsim, the MI355X-native vectorised DexHand simulator core.
 *
 * This is the drop-in boundary underneath the reference's Python env surface
 * (reference: dexhand_env/factory.py:93-147 make_env, dexhand_env/tasks/dexhand_base.py:893-942 step,
 * :805-838 reset, :743-803 reset_idx).  The reference has no FFI of its own; every entry
 * point below replaces one *opaque Isaac Gym call site* or one *Python component call* on that path and
 * cites it.  Plain pointers and sizes only: no torch types, no exceptions, int error codes.
 *
 * Memory model: the caller (PyTorch-ROCm host side) allocates device memory and hands pointers in.
 *   - one "arena" of 4-byte words holding every per-env SoA field as [rows][num_envs]
 *     (layout is computed by dexsim_arena_layout and is the *state-buffer contract* that replaces
 *     reference components/physics/tensor_manager.py:139-376);
 *   - a handful of AoS "API tensors" with the reference's own shapes (dof_state (N,26,2), root (N,A,13),
 *     obs_buf (N,O), rew (N), reset (N) u8, episode_step_count (N) i64, ...).
 * The library never allocates or frees per-env memory and never synchronises the stream.
 */
#ifndef DEXSIM_H
#define DEXSIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ fixed topology of the DexHand chain */
#define DEXSIM_NJ        26   /* DOFs: 6 base + 20 finger (reference constants.py:8-11)                 */
#define DEXSIM_NBASE      6
#define DEXSIM_NFINGER    5
#define DEXSIM_NFJ        4   /* joints per finger                                                      */
#define DEXSIM_NACT      18   /* active targets: 6 base + 12 finger controls                            */
#define DEXSIM_NSITE     11   /* right_hand_base, 5 tips, 5 pads (hand_initializer.py:120-134,499-501)  */
#define DEXSIM_NCAP      18   /* collision capsules: 3 palm + 3 per finger                              */
#define DEXSIM_NFSLOT    17   /* net-contact-force slots: 15 finger links, palm, box                    */
#define DEXSIM_FSLOT_PALM 15
#define DEXSIM_FSLOT_BOX  16
#define DEXSIM_KMAX      24   /* max active contacts per env per sub-step (<= 4 box/ground + hand contacts in priority order) */
#define DEXSIM_BISECT_ITERS 10 /* capsule / box narrowphase: bisection steps for the point of the capsule axis nearest to the box
                                (2^-10 of the axis, ~30 um; the gap is stationary there: its error is second order) */
#define DEXSIM_NWKEY 128     /* warm-start cache slots of the contact solver: (capsule * 2 + type) * 2 + sample for hand contacts (< 72),
                                80 + list slot for the (<= 4) box/ground contacts (their tag carries the box corner),
                                88 + (finger joint * 2 + side) for the joint-limit rows (side 0 = lower, 1 = upper) */
/* the warm-start generation of an env advances once per sub-step and once per reset; it wraps at 2^27 so that the cache tags
   8 * generation + corner stay below 2^30 in signed 32-bit arithmetic for any run length (a generation is only ever compared
   with the one before it: a tag can falsely match only if its slot sat unwritten for exactly 2^27 generations) */
#define DEXSIM_WGEN_NEXT(g) (((g) + 1) & 0x07ffffff)
#define DEXSIM_NRESET_SAMPLES 29 /* rand draws of one reset (blind_grasping_task.py:449-547)            */
#define DEXSIM_MAX_OBS_SEG 40

/* task kinds (factory.py:46-62) */
#define DEXSIM_TASK_BASE           0
#define DEXSIM_TASK_BLIND_GRASPING 1

/* control modes (action_processor.py:197-199) */
#define DEXSIM_MODE_POSITION       0
#define DEXSIM_MODE_POSITION_DELTA 1

/* error codes */
#define DEXSIM_OK              0
#define DEXSIM_ERR_ARG         1
#define DEXSIM_ERR_NOT_BOUND   2
#define DEXSIM_ERR_HIP         3
#define DEXSIM_ERR_NO_DEVICE   4
#define DEXSIM_ERR_LAYOUT      5

/* ------------------------------------------------------------------ observation dictionary layout
 * One row block per obs_dict key of the reference (observation_encoder.py:576-758,
 * blind_grasping_task.py:549-653).  obs_all is SoA [DEXSIM_OBS_ALL_DIM][N]; obs_buf (N,O) is the
 * concatenation of the configured policy_observation_keys (observation_encoder.py:783-829). */
#define DEXSIM_OBS_KEYS(X) \
  X(base_dof_pos, 6) X(base_dof_vel, 6) X(active_finger_dof_pos, 12) X(active_finger_dof_vel, 12) \
  X(all_finger_dof_pos, 20) X(all_finger_dof_vel, 20) X(hand_pose, 7) X(hand_pose_arr_aligned, 7) \
  X(contact_forces, 15) X(prev_actions, 18) X(active_prev_targets, 18) X(base_dof_target, 6) \
  X(active_finger_dof_target, 12) X(all_finger_dof_target, 20) X(contact_force_magnitude, 5) \
  X(contact_binary, 5) X(contact_duration, 5) X(fingertip_poses_world, 35) X(fingertip_poses_hand, 35) \
  X(fingerpad_poses_world, 35) X(fingerpad_poses_hand, 35) X(episode_time, 1) X(active_rule_targets, 18) \
  X(object_pos, 3) X(object_vel, 3) X(finger_to_object_distances, 5) X(avg_finger_to_object_distance, 1) \
  X(finger_to_object_height_diff, 5) X(avg_finger_to_object_height_diff, 1) X(hand_to_object_distance, 1) \
  X(fingerpad_distances, 10) X(first_three_fingerpad_centroid, 3) X(thumb_contact, 1) \
  X(other_fingers_contact, 1) X(grasp_state, 1) X(grasp_duration, 1) X(current_stage, 1) \
  X(time_in_stage, 1) X(stage_progress, 1)

enum {
#define X(name, dim) DEXSIM_OBSKEY_##name,
  DEXSIM_OBS_KEYS(X)
#undef X
  DEXSIM_NUM_OBS_KEYS
};
#define DEXSIM_OBS_ALL_DIM 392
#define DEXSIM_OBS_BASE_TASK_DIM 353 /* keys before object_pos exist for every task */

/* ------------------------------------------------------------------ reward terms
 * common: reward_calculator.py:66-221 ; task: blind_grasping_task.py:980-1208 */
#define DEXSIM_REWARD_TERMS(X) \
  X(alive) X(height_safety) X(finger_velocity) X(hand_velocity) X(hand_angular_velocity) X(joint_limit) \
  X(finger_acceleration) X(hand_acceleration) X(hand_angular_acceleration) X(contact_stability) \
  X(s1_height_alignment) X(s1_centroid_positioning) X(s1_object_stability) X(s1_finger_height_consistency) \
  X(s1_thumb_rotation) X(s2_thumb_contact) X(s2_other_fingers_contact) X(s2_grasp_achievement) \
  X(s2_fingerpad_proximity) X(s2_base_stability) X(s3_object_height) X(s3_grasp_maintenance) \
  X(s3_grasp_duration) X(s1_completion) X(s2_completion) X(penetration_penalty)

enum {
#define X(name) DEXSIM_REW_##name,
  DEXSIM_REWARD_TERMS(X)
#undef X
  DEXSIM_NUM_REWARD_TERMS
};
#define DEXSIM_NUM_COMMON_REWARD_TERMS 10
/* reward component rows: [0,26) raw, [26,52) weighted, 52 total, 53..55 termination raw
 * (success, failure_penalty, timeout_penalty), 56..58 termination weighted (step_processor.py:204-219) */
#define DEXSIM_REWROW_WEIGHTED 26
#define DEXSIM_REWROW_TOTAL    52
#define DEXSIM_REWROW_TERM_RAW 53
#define DEXSIM_REWROW_TERM_W   56
#define DEXSIM_NUM_REWROWS     59

/* termination criteria (step_processor.py:133-181, blind_grasping_task.py:1210-1332) */
#define DEXSIM_SUCC_grasp_lift_success 0
#define DEXSIM_NUM_SUCC 1
#define DEXSIM_FAIL_hitting_ground         0
#define DEXSIM_FAIL_box_too_far            1
#define DEXSIM_FAIL_stage1_pregrasp_failed 2
#define DEXSIM_FAIL_stage2_contact_failed  3
#define DEXSIM_FAIL_stage3_grasp_lost      4
#define DEXSIM_NUM_FAIL 5
#define DEXSIM_MASK_SUCCESS 0
#define DEXSIM_MASK_FAILURE 1
#define DEXSIM_MASK_TIMEOUT 2
#define DEXSIM_MASK_SUCC_REASON 3                       /* [NUM_SUCC] */
#define DEXSIM_MASK_FAIL_REASON (3 + DEXSIM_NUM_SUCC)   /* [NUM_FAIL] */
#define DEXSIM_NUM_MASKS (3 + DEXSIM_NUM_SUCC + DEXSIM_NUM_FAIL)

/* global statistics block (device, 64 floats): written once per step by the finalize kernel.
 * termination_manager.py:164-185,259-266 (means / rates), :323-339 (consecutive successes). */
#define DEXSIM_STAT_SUCC_MEAN   0  /* [NUM_SUCC] mean of each success criterion                */
#define DEXSIM_STAT_FAIL_MEAN   4  /* [NUM_FAIL] mean of each failure criterion                */
#define DEXSIM_STAT_SUCCESS_RATE 12
#define DEXSIM_STAT_FAILURE_RATE 13
#define DEXSIM_STAT_TIMEOUT_RATE 14
#define DEXSIM_STAT_CONSECUTIVE_SUCCESSES 15
#define DEXSIM_STAT_NUM_RESETS  16 /* envs reset in this control step                          */
#define DEXSIM_STAT_PHYSICS_STEPS 17 /* physics steps executed in this control step (1 or 2)   */
#define DEXSIM_STAT_MEAN_CONTACTS 18 /* mean active contacts/env in the last sub-step of the step's main physics step */
#define DEXSIM_STAT_MEAN_HAND_CONTACTS 19 /* ... of which hand/box and hand/ground (the rest is box/ground)            */
#define DEXSIM_STAT_USED  20 /* statistics words in use (the rest of the block is reserved)            */
#define DEXSIM_STAT_WORDS 64

/* number of hand bodies published in rigid_body_states: 7 base-chain + 30 finger bodies */
#define DEXSIM_NUM_HAND_BODIES 37

/* ------------------------------------------------------------------ hand model (authored stand-in for the
 * absent MJCF dexhand021_right_simplified_floating.xml, dexhand_base.py:171).
 * Joint j: frame = parent_frame * Trans(poff) * Rot(qoff) * Motion(axis, q); quaternions are xyzw.  Joint 0's parent is the
 * spawn pose.  Finger f (0..4) owns joints 6+4f .. 9+4f, chained; joint 6+4f's parent is joint 5. */
typedef struct DexHandModel {
  float spawn_pos[3];
  float spawn_quat[4];              /* xyzw */
  int   jtype[DEXSIM_NJ];           /* 0 prismatic, 1 revolute */
  float jqoff[DEXSIM_NJ][4];        /* xyzw */
  float jpoff[DEXSIM_NJ][3];
  float jaxis[DEXSIM_NJ][3];        /* unit, in the joint's own frame */
  float mass[DEXSIM_NJ];            /* body carried by joint j */
  float com[DEXSIM_NJ][3];          /* in joint frame */
  float inertia[DEXSIM_NJ][6];      /* about COM, joint-frame axes: xx yy zz xy xz yz */
  float kp[DEXSIM_NJ], kd[DEXSIM_NJ], armature[DEXSIM_NJ], lo[DEXSIM_NJ], hi[DEXSIM_NJ];
  int   site_parent[DEXSIM_NSITE];  /* joint index the site is welded to */
  float site_q[DEXSIM_NSITE][4];
  float site_p[DEXSIM_NSITE][3];
  int   cap_parent[DEXSIM_NCAP];
  float cap_p0[DEXSIM_NCAP][3];
  float cap_p1[DEXSIM_NCAP][3];
  float cap_r[DEXSIM_NCAP];
  int   cap_fslot[DEXSIM_NCAP];     /* net-contact-force slot of the owning body */
  float hand_friction;
  /* published rigid bodies of the hand actor (rigid_body_states rows): welded to a joint frame
   * (-1 = the fixed spawn frame) by a constant transform */
  int   body_parent[DEXSIM_NUM_HAND_BODIES];
  float body_q[DEXSIM_NUM_HAND_BODIES][4];
  float body_p[DEXSIM_NUM_HAND_BODIES][3];
  int   body_fslot[DEXSIM_NUM_HAND_BODIES]; /* net-contact-force slot or -1 */
} DexHandModel;

/* ------------------------------------------------------------------ resolved configuration
 * (the cfg dict keys the reference consumes, SURVEY.md §8b "cfg keys consumed") */
typedef struct DexSimConfig {
  int   num_envs;
  int   task;                       /* DEXSIM_TASK_* */
  /* sim.* (vec_task.py:226-296, cfg/physics/default.yaml) */
  float dt;
  int   substeps;
  float gravity[3];
  int   num_position_iterations;
  float contact_offset, rest_offset, max_depenetration_velocity;
  float erp;                        /* penetration recovery fraction per sub-step (build's choice) */
  float control_dt;                 /* 2*dt: dexhand_base.py:270-320, physics_manager.py:243-270 */
  int   episode_length;             /* env.episodeLength */
  uint32_t seed;                    /* train.seed */
  /* action (action_processor.py:181-247, 380-434) */
  int   control_mode;
  int   policy_controls_base, policy_controls_fingers;
  int   num_actions;
  float max_deltas[DEXSIM_NACT];
  float active_lower[DEXSIM_NACT], active_upper[DEXSIM_NACT];
  /* observation */
  float contact_binary_threshold;
  int   num_obs;
  int   n_obs_seg;
  int   obs_seg_off[DEXSIM_MAX_OBS_SEG];   /* row offset into obs_all */
  int   obs_seg_len[DEXSIM_MAX_OBS_SEG];
  /* termination (termination_manager.py:26-75) */
  int   height_safety_enabled;
  float handbase_threshold, fingertip_threshold;
  int   active_success_mask;        /* bit i = success criterion i active */
  int   active_failure_mask;
  float success_reward, failure_penalty, timeout_penalty;
  int   max_consecutive_successes;
  /* reward weights (reward_weights section of the task YAML) */
  float reward_weight[DEXSIM_NUM_REWARD_TERMS];
  /* ground + box (dexhand_base.py:632-638, blind_grasping_task.py:138-144) */
  float ground_friction;
  int   has_box;
  float box_size, box_mass, box_friction, box_xy_range, box_z;
  /* BlindGrasping task params (blind_grasping_task.py:146-199) */
  float height_threshold, contact_duration_threshold_s;
  int   contact_duration_threshold_steps;  /* int(threshold / control_dt), :271-273 */
  int   min_fingers_for_grasp;
  float max_box_distance, stage1_duration, stage2_duration;
  float hand_translation_range, hand_rotation_range, thumb_rotation_range, other_finger_range;
  float stage2_contact_success_threshold;
  float height_alignment_decay, centroid_positioning_decay, object_stability_decay;
  float first_three_height_consistency_decay, fingerpad_proximity_decay, base_stability_decay;
  float geometric_penetration_factor, proximity_min_distance_factor, penetration_depth_scale;
  float height_tolerance, centroid_tolerance, position_drift_tolerance, velocity_tolerance;
  /* per-env domain randomisation of the box (BASELINE config #5; new capability) */
  int   dr_enabled;
  float dr_mass_lo, dr_mass_hi, dr_mu_lo, dr_mu_hi;
  uint32_t dr_seed;
  /* static box actor (the harness's contact-test box, examples/dexhand_test.py:950-1024: gym.create_box with
   * fix_base_link = True at a fixed pose, identity orientation).  With box_fixed the box never moves: no gravity, no
   * box/ground contacts, infinite mass in the hand/box contact rows; resets leave it where it is.  Works with either task
   * (BaseTask has no box of its own: has_box = 1 then adds the second actor). */
  int   box_fixed;
  float box_fixed_pos[3];
  /* Joint limits as unilateral solver rows (round 3; the reference's limits are PhysX articulation limits, dof_props lower / upper,
   * tensor_manager.py:547-554).  With joint_limit_rows != 0 every finger joint within joint_limit_margin (rad) of a limit gets a
   * one-row speculative constraint (J = +-e_j, gap = distance to the limit, no friction) in the contact solver -- for the fingers
   * that have at least one contact in the sub-step: without contact forces the PD drive cannot push a joint across a limit its
   * own target respects, and the position clamp behind the integration stays as the safety net in every case.  List order: a
   * finger's limit rows follow that finger's contacts; they count against DEXSIM_KMAX; each is a solver block of its own. */
  int   joint_limit_rows;
  float joint_limit_margin;
} DexSimConfig;

/* ------------------------------------------------------------------ arena layout */
typedef struct DexSimField {
  char   name[40];
  int    rows;       /* field occupies rows*num_envs words at offset */
  int    is_int;     /* 1: int32 words, 0: float32 */
  size_t offset;     /* in 4-byte words from arena base */
} DexSimField;

/* API tensors (AoS, reference shapes).  All device pointers; any may be NULL except obs_buf/rew/reset. */
typedef struct DexSimBuffers {
  void*    arena;              /* arena_words * 4 bytes                                             */
  float*   stats;              /* DEXSIM_STAT_WORDS floats                                          */
  int*     counters;           /* DEXSIM_STAT_WORDS ints, scratch for the statistics reduction      */
  float*   obs_buf;            /* (N, num_obs) row-major  -- DexHandBase.obs_buf                    */
  float*   rew_buf;            /* (N,)                                                              */
  uint8_t* reset_buf;          /* (N,) bool                                                         */
  int64_t* episode_step_count; /* (N,) int64 (initialization_manager.py:47-49)                      */
  int64_t* episode_length;     /* (N,) int64 extras["episode_length"] (step_processor.py:221-232)   */
  float*   dof_state;          /* (N, 26, 2)  gym.refresh_dof_state_tensor                          */
  float*   root_state;         /* (N, A, 13)  gym.refresh_actor_root_state_tensor; A = 1 + has_box  */
  float*   rigid_body_states;  /* (N, B, 13)  filled by dexsim_refresh_body_states only             */
  float*   contact_forces_all; /* (N, B, 3)   filled by dexsim_refresh_body_states only             */
  float*   full_dof_targets;   /* (N, 26)     ActionProcessor.full_dof_targets                      */
  float*   reset_samples;      /* (N, 29) uniforms in [0,1) or NULL -> device Philox stream         */
  uint8_t* masks;              /* (DEXSIM_NUM_MASKS, N) bool rows, extras masks (step_processor.py:221-232,
                                  termination_manager.py:246-256): success, failure, timeout,
                                  success_reason[NUM_SUCC], failure_reason[NUM_FAIL]; may be NULL     */
  float*   raw_targets;        /* (N, 18) or NULL: output of a host-side custom action rule
                                  (ActionRules.set_action_rule, rules.py:211-224) replacing the built-in
                                  position / position_delta rule; filters + coupling still run in-kernel */
} DexSimBuffers;

typedef struct DexSim* dexsim_t;

/* sizeof() of {DexHandModel, DexSimConfig, DexSimField, DexSimBuffers}: lets a foreign-language
 * binding verify its struct mirrors before passing pointers. */
int dexsim_struct_sizes(size_t out[4]);

/* Layout of the arena for this configuration.  Replaces TensorManager.acquire_tensor_handles
 * (tensor_manager.py:80-136): tells the host what to allocate and where each field lives. */
int dexsim_arena_layout(const DexSimConfig* cfg, DexSimField* fields, int max_fields, int* n_fields,
                        size_t* arena_words);

/* obs_dict key table: name / row offset / dim of key i (0 <= i < DEXSIM_NUM_OBS_KEYS). */
int dexsim_obs_key_info(int i, const char** name, int* offset, int* dim);
/* reward term name i (0 <= i < DEXSIM_NUM_REWARD_TERMS). */
int dexsim_reward_term_name(int i, const char** name);
/* rigid body name i of the hand actor (0 <= i < DEXSIM_NUM_HAND_BODIES). */
int dexsim_body_name(int i, const char** name);

/* create_sim + load_asset + create_actor for every env (vec_task.py:298-311,
 * hand_initializer.py:209-257,365-422, blind_grasping_task.py:300-366). No per-env allocation. */
int dexsim_create(const DexSimConfig* cfg, const DexHandModel* model, int device, dexsim_t* out);
int dexsim_destroy(dexsim_t h);

/* gym.acquire_*_tensor + gymtorch.wrap_tensor + gym.prepare_sim (tensor_manager.py:99-136,
 * dexhand_base.py:431): bind caller-owned device memory. */
int dexsim_bind(dexsim_t h, const DexSimBuffers* bufs);

/* Zero all state, put the hand at the spawn pose, box at its default pose, FSM at stage 1, draw the
 * per-env DR parameters.  (scene construction, dexhand_base.py:609-672.) */
int dexsim_init_state(dexsim_t h, void* stream);

/* ActionProcessor.process_actions + gym.set_dof_position_target_tensor
 * (action_processor.py:284-352); also ObservationEncoder.update_prev_actions (:287-296).
 * actions: (N, num_actions) device.  zero_targets != 0 reproduces the pre-finalize_setup branch
 * (:305-318). */
int dexsim_process_actions(dexsim_t h, const float* actions, int zero_targets, void* stream);

/* Opens a control step whose action stage ran on the host (custom post-action filters / coupling rule,
 * action_processor.py:698-720, write targets and active_prev_targets into the arena directly): what
 * dexsim_process_actions does for the device-side step bookkeeping (new stamp of the device-side reset gate, contact
 * statistics words), without touching targets. */
int dexsim_begin_step(dexsim_t h, void* stream);

/* PhysicsManager.step_physics = gym.simulate + fetch_results + 4 refreshes
 * (physics_manager.py:73-119): one sim.dt = `substeps` sub-steps for every env, then publish
 * dof_state/root_state and the compact L1 state.  gate_on_reset != 0 makes every kernel a no-op unless
 * the device-side "some env reset this step" flag is set (reset_manager.py:180 without a host sync). */
int dexsim_physics_step(dexsim_t h, int gate_on_reset, void* stream);

/* StepProcessor.process_physics_step (step_processor.py:37-131) up to and including the reset of
 * finished envs and the extra physics step.  obs_only != 0 = compute_observations +
 * concatenate_observations only (dexhand_base.py:811-834). */
int dexsim_post_physics(dexsim_t h, int obs_only, void* stream);

/* DexHandBase.step (dexhand_base.py:893-942): process_actions, physics, post-physics, in-step resets,
 * conditional extra physics step, extras statistics.  No host synchronisation.  With substeps == 4 this is two
 * launches: one kernel carrying the whole control step up to the reset gate, and its device-gated twin (the extra
 * physics step + reset phase 1 + statistics); other sub-step counts go through the three staged calls above. */
int dexsim_step(dexsim_t h, const float* actions, void* stream);

/* DexHandBase.reset_idx (dexhand_base.py:743-803) for caller-chosen envs: env_ids is a device array
 * of k int64.  Includes the unconditional physics step of ResetManager.reset_idx (reset_manager.py:180). */
int dexsim_reset_idx(dexsim_t h, const int64_t* env_ids, int k, void* stream);

/* DexHandBase.reset (dexhand_base.py:805-838): reset_idx(all) + observations + post_physics_step. */
int dexsim_reset(dexsim_t h, void* stream);

/* gym.refresh_rigid_body_state_tensor + refresh_net_contact_force_tensor for the full (N,B,*)
 * tensors (physics_manager.py:108-109), materialised on demand. */
int dexsim_refresh_body_states(dexsim_t h, void* stream);

/* gym.set_dof_state_tensor_indexed / set_actor_root_state_tensor_indexed
 * (physics_manager.py:146-151, reset_manager.py:153-158): ingest the AoS API tensors for k envs.
 * (The contact solver's warm-start cache -- arena field `wlam`, generation `wgen` -- is invalidated by resets, not by these
 * setters: after a teleport a stale entry is only the starting guess of the first sub-step's solve.) */
int dexsim_set_dof_state_indexed(dexsim_t h, const int64_t* env_ids, int k, void* stream);
int dexsim_set_root_state_indexed(dexsim_t h, const int64_t* env_ids, int k, void* stream);

/* Rollout sink (new capability, SURVEY.md 8e / 8f-1: the "PPO buffer" end of the path): besides obs_buf / rew_buf /
 * reset_buf the post-physics flush of every following dexsim_step / dexsim_post_physics also writes the step's
 * observations (N, num_obs) f32, rewards (N) f32 and done flags (N) u8 to these device pointers -- typically row t of
 * the caller's (T, N, ...) rollout tensors -- so that collecting a rollout needs no copy kernels.  Any pointer may be
 * NULL; all NULL switches the sink off.  The pointers are kernel arguments: no device memory is touched by this call. */
int dexsim_set_step_sink(dexsim_t h, float* obs, float* rew, uint8_t* done);

/* Statistics sink: the launch that closes a control step (finalize_stats: the cross-env means / rates of
 * TerminationManager.evaluate, termination_manager.py:160-185, and consecutive successes, :323-339) also copies the first
 * DEXSIM_STAT_USED words of the statistics block to `dst` -- row t of the caller's (T, DEXSIM_STAT_WORDS) rollout tensor -- so that
 * envs sharded over ranks can reduce their whole-population statistics with ONE small all-reduce per rollout (rollout.py:
 * reduce_stats) instead of one per step.  NULL switches it off.  Kernel argument like the step sink. */
int dexsim_set_stats_sink(dexsim_t h, float* dst);

/* get_observations_dict() (dexhand_base.py:948-956) is served from the arena field obs_all: 392 SoA rows per env that the post
 * block writes in every step (mode 0, the default: every key of the reference's obs_dict is a view).  A training loop reads
 * obs_buf only -- the configured policy keys -- so mode 1 ("policy") stops materialising the rows: 1 568 B per env and step less
 * to write; the host side then serves the policy keys as views of obs_buf and raises on any other key.  Kernel argument. */
int dexsim_set_obs_dict_mode(dexsim_t h, int mode);

/* Phase probe (measurement, SURVEY.md 8d: the contact-solve sub-metric on the PRODUCTION kernel): with `buf` set --
 * 4 x ceil(num_envs / 64) uint32 on the device, zeroed by the caller -- every following physics launch of the 4-sub-step path
 * adds, per workgroup, shader-clock ticks (s_memtime) to buf[4 w + 0] = phases 3 + 4 of the general contact path (contact rows +
 * the block solver's sweeps), [4 w + 1] = the whole launch (ungated launches), [4 w + 3] = phase 4 alone, and counts in
 * [4 w + 2] the sub-steps that ran the general path.  Solver time = launch time (HIP events) x buf[0] / buf[1].  NULL switches it
 * off (the default; the probe then costs two scalar branches per sub-step). */
int dexsim_set_phase_probe(dexsim_t h, uint32_t* buf);

/* DexHandBase.pre_physics_step keeps `self.actions = actions.clone()` (dexhand_base.py:851).  With a destination set here
 * ((N, num_actions) f32 on the device, or NULL to switch off) the action block writes that copy itself, so the host side
 * needs no clone kernel per step.  Kernel argument like the step sink. */
int dexsim_set_action_copy(dexsim_t h, float* dst);

/* Test / profiling hooks: run one pipeline stage on the bound buffers. */
#define DEXSIM_STAGE_DYNAMICS 0  /* FK + CRBA + bias + factorisation + narrowphase + row build (stand-alone kernel) */
#define DEXSIM_STAGE_SOLVE    1  /* PGS contact-impulse solve + integrate (stand-alone kernel)      */
#define DEXSIM_STAGE_PUBLISH  2  /* FK of sites, AoS dof_state/root publication                     */
#define DEXSIM_STAGE_POST     3  /* fused obs + FSM + termination + reward                          */
#define DEXSIM_STAGE_RESET    4  /* masked reset of envs whose reset_buf is set                     */
#define DEXSIM_STAGE_FINALIZE 5  /* statistics                                                      */
#define DEXSIM_STAGE_SUBSTEP  6  /* one sub-step (DYNAMICS + SOLVE + integration + publication) as its own launch */
#define DEXSIM_STAGE_PHYSICS  7  /* physics step alone: all `substeps` of a sim.dt (one launch when substeps == 4)      */
#define DEXSIM_STAGE_STEP     8  /* the launch dexsim_step issues: actions + 4 sub-steps + post-physics (re-uses the    */
                                 /* action pointer of the last dexsim_step; advances the simulation)                    */
int dexsim_run_stage(dexsim_t h, int stage, void* stream);

/* Time `launches` launches of one stage, each bracketed by a hipEvent pair and a host synchronisation, and return the
 * mean duration in microseconds.  The event fences make every launch start from a cold L2, so these are upper bounds
 * (bench.py uses them for the stand-alone kernels only). */
int dexsim_time_stage(dexsim_t h, int stage, int launches, void* stream, float* mean_us);

/* In-situ timing of the main launch of dexsim_step (k_physics4 with the action and post-physics blocks when
 * substeps == 4): enable != 0 starts recording a hipEvent pair around that launch on every following dexsim_step
 * (ring of 64, no host synchronisation, so the launches stay back to back as in production); enable == 0 stops,
 * synchronises and returns the mean duration in microseconds over the *n recorded steps.  (The event fences still
 * cost the kernel its warm L2: +30 % on MI355X; bench.py therefore times the whole region instead.) */
int dexsim_step_timing(dexsim_t h, int enable, float* mean_us, int* n);

const char* dexsim_error_string(int code);
const char* dexsim_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* DEXSIM_H */
