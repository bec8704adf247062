"""ctypes mirrors of the structs in include/dexsim.h (the C-ABI drop-in boundary).

Kept in lock-step with the header; `check_struct_sizes` verifies sizeof() against the loaded library so
a drifted mirror fails loudly instead of corrupting memory.
"""
import ctypes as C

NJ, NBASE, NFINGER, NFJ, NACT = 26, 6, 5, 4, 18
NSITE, NCAP, NFSLOT, KMAX = 11, 18, 17, 24
NWKEY = 128     # DEXSIM_NWKEY: warm-start cache slots per env
FSLOT_PALM, FSLOT_BOX = 15, 16
NRESET_SAMPLES = 29
MAX_OBS_SEG = 40
NUM_HAND_BODIES = 37
OBS_ALL_DIM = 392
NUM_REWARD_TERMS = 26
NUM_COMMON_REWARD_TERMS = 10
REWROW_WEIGHTED, REWROW_TOTAL, REWROW_TERM_RAW, REWROW_TERM_W, NUM_REWROWS = 26, 52, 53, 56, 59
NUM_SUCC, NUM_FAIL = 1, 5
NUM_MASKS = 3 + NUM_SUCC + NUM_FAIL
MASK_SUCCESS, MASK_FAILURE, MASK_TIMEOUT, MASK_SUCC_REASON, MASK_FAIL_REASON = 0, 1, 2, 3, 3 + NUM_SUCC
STAT_WORDS = 64
STAT_USED = 20
STAT = dict(SUCC_MEAN=0, FAIL_MEAN=4, SUCCESS_RATE=12, FAILURE_RATE=13, TIMEOUT_RATE=14,
            CONSECUTIVE_SUCCESSES=15, NUM_RESETS=16, PHYSICS_STEPS=17, MEAN_CONTACTS=18, MEAN_HAND_CONTACTS=19)

TASK_BASE, TASK_BLIND_GRASPING = 0, 1
MODE_POSITION, MODE_POSITION_DELTA = 0, 1
STAGE = dict(DYNAMICS=0, SOLVE=1, PUBLISH=2, POST=3, RESET=4, FINALIZE=5, SUBSTEP=6, PHYSICS=7, STEP=8)

SUCCESS_CRITERIA = ["grasp_lift_success"]
FAILURE_CRITERIA = ["hitting_ground", "box_too_far", "stage1_pregrasp_failed",
                    "stage2_contact_failed", "stage3_grasp_lost"]

f32, i32, u32 = C.c_float, C.c_int, C.c_uint32


class DexHandModel(C.Structure):
    _fields_ = [
        ("spawn_pos", f32 * 3), ("spawn_quat", f32 * 4),
        ("jtype", i32 * NJ),
        ("jqoff", (f32 * 4) * NJ), ("jpoff", (f32 * 3) * NJ), ("jaxis", (f32 * 3) * NJ),
        ("mass", f32 * NJ), ("com", (f32 * 3) * NJ), ("inertia", (f32 * 6) * NJ),
        ("kp", f32 * NJ), ("kd", f32 * NJ), ("armature", f32 * NJ), ("lo", f32 * NJ), ("hi", f32 * NJ),
        ("site_parent", i32 * NSITE), ("site_q", (f32 * 4) * NSITE), ("site_p", (f32 * 3) * NSITE),
        ("cap_parent", i32 * NCAP), ("cap_p0", (f32 * 3) * NCAP), ("cap_p1", (f32 * 3) * NCAP),
        ("cap_r", f32 * NCAP), ("cap_fslot", i32 * NCAP),
        ("hand_friction", f32),
        ("body_parent", i32 * NUM_HAND_BODIES), ("body_q", (f32 * 4) * NUM_HAND_BODIES),
        ("body_p", (f32 * 3) * NUM_HAND_BODIES), ("body_fslot", i32 * NUM_HAND_BODIES),
    ]


class DexSimConfig(C.Structure):
    _fields_ = [
        ("num_envs", i32), ("task", i32),
        ("dt", f32), ("substeps", i32), ("gravity", f32 * 3),
        ("num_position_iterations", i32),
        ("contact_offset", f32), ("rest_offset", f32), ("max_depenetration_velocity", f32),
        ("erp", f32), ("control_dt", f32), ("episode_length", i32), ("seed", u32),
        ("control_mode", i32), ("policy_controls_base", i32), ("policy_controls_fingers", i32),
        ("num_actions", i32),
        ("max_deltas", f32 * NACT), ("active_lower", f32 * NACT), ("active_upper", f32 * NACT),
        ("contact_binary_threshold", f32), ("num_obs", i32), ("n_obs_seg", i32),
        ("obs_seg_off", i32 * MAX_OBS_SEG), ("obs_seg_len", i32 * MAX_OBS_SEG),
        ("height_safety_enabled", i32), ("handbase_threshold", f32), ("fingertip_threshold", f32),
        ("active_success_mask", i32), ("active_failure_mask", i32),
        ("success_reward", f32), ("failure_penalty", f32), ("timeout_penalty", f32),
        ("max_consecutive_successes", i32),
        ("reward_weight", f32 * NUM_REWARD_TERMS),
        ("ground_friction", f32), ("has_box", i32),
        ("box_size", f32), ("box_mass", f32), ("box_friction", f32), ("box_xy_range", f32), ("box_z", f32),
        ("height_threshold", f32), ("contact_duration_threshold_s", f32),
        ("contact_duration_threshold_steps", i32), ("min_fingers_for_grasp", i32),
        ("max_box_distance", f32), ("stage1_duration", f32), ("stage2_duration", f32),
        ("hand_translation_range", f32), ("hand_rotation_range", f32),
        ("thumb_rotation_range", f32), ("other_finger_range", f32),
        ("stage2_contact_success_threshold", f32),
        ("height_alignment_decay", f32), ("centroid_positioning_decay", f32),
        ("object_stability_decay", f32), ("first_three_height_consistency_decay", f32),
        ("fingerpad_proximity_decay", f32), ("base_stability_decay", f32),
        ("geometric_penetration_factor", f32), ("proximity_min_distance_factor", f32),
        ("penetration_depth_scale", f32),
        ("height_tolerance", f32), ("centroid_tolerance", f32),
        ("position_drift_tolerance", f32), ("velocity_tolerance", f32),
        ("dr_enabled", i32), ("dr_mass_lo", f32), ("dr_mass_hi", f32), ("dr_mu_lo", f32), ("dr_mu_hi", f32),
        ("dr_seed", u32),
        ("box_fixed", i32), ("box_fixed_pos", f32 * 3),
        ("joint_limit_rows", i32), ("joint_limit_margin", f32),
    ]


class DexSimField(C.Structure):
    _fields_ = [("name", C.c_char * 40), ("rows", i32), ("is_int", i32), ("offset", C.c_size_t)]


class DexSimBuffers(C.Structure):
    _fields_ = [
        ("arena", C.c_void_p), ("stats", C.c_void_p), ("counters", C.c_void_p),
        ("obs_buf", C.c_void_p), ("rew_buf", C.c_void_p), ("reset_buf", C.c_void_p),
        ("episode_step_count", C.c_void_p), ("episode_length", C.c_void_p),
        ("dof_state", C.c_void_p), ("root_state", C.c_void_p),
        ("rigid_body_states", C.c_void_p), ("contact_forces_all", C.c_void_p),
        ("full_dof_targets", C.c_void_p), ("reset_samples", C.c_void_p),
        ("masks", C.c_void_p), ("raw_targets", C.c_void_p),
    ]


# every symbol include/dexsim.h declares (checked by tests/test_abi.py against the built library)
EXPORTED_SYMBOLS = [
    "dexsim_struct_sizes", "dexsim_arena_layout", "dexsim_obs_key_info", "dexsim_reward_term_name",
    "dexsim_body_name", "dexsim_create", "dexsim_destroy", "dexsim_bind", "dexsim_init_state",
    "dexsim_process_actions", "dexsim_begin_step", "dexsim_physics_step", "dexsim_post_physics", "dexsim_step",
    "dexsim_reset_idx", "dexsim_reset", "dexsim_refresh_body_states", "dexsim_set_dof_state_indexed",
    "dexsim_set_root_state_indexed", "dexsim_run_stage", "dexsim_time_stage", "dexsim_step_timing", "dexsim_set_step_sink", "dexsim_set_stats_sink", "dexsim_set_phase_probe", "dexsim_set_obs_dict_mode", "dexsim_set_action_copy", "dexsim_error_string",
    "dexsim_last_error",
]


def declare_prototypes(lib):
    """Attach argtypes/restype for every entry point of include/dexsim.h to a loaded CDLL."""
    P = C.POINTER
    vp, sz = C.c_void_p, C.c_size_t
    lib.dexsim_struct_sizes.argtypes = [P(sz)]
    lib.dexsim_arena_layout.argtypes = [P(DexSimConfig), P(DexSimField), i32, P(i32), P(sz)]
    lib.dexsim_obs_key_info.argtypes = [i32, P(C.c_char_p), P(i32), P(i32)]
    lib.dexsim_reward_term_name.argtypes = [i32, P(C.c_char_p)]
    lib.dexsim_body_name.argtypes = [i32, P(C.c_char_p)]
    lib.dexsim_create.argtypes = [P(DexSimConfig), P(DexHandModel), i32, P(vp)]
    lib.dexsim_destroy.argtypes = [vp]
    lib.dexsim_bind.argtypes = [vp, P(DexSimBuffers)]
    lib.dexsim_init_state.argtypes = [vp, vp]
    lib.dexsim_process_actions.argtypes = [vp, vp, i32, vp]
    lib.dexsim_begin_step.argtypes = [vp, vp]
    lib.dexsim_physics_step.argtypes = [vp, i32, vp]
    lib.dexsim_post_physics.argtypes = [vp, i32, vp]
    lib.dexsim_step.argtypes = [vp, vp, vp]
    lib.dexsim_reset_idx.argtypes = [vp, vp, i32, vp]
    lib.dexsim_reset.argtypes = [vp, vp]
    lib.dexsim_refresh_body_states.argtypes = [vp, vp]
    lib.dexsim_set_dof_state_indexed.argtypes = [vp, vp, i32, vp]
    lib.dexsim_set_root_state_indexed.argtypes = [vp, vp, i32, vp]
    lib.dexsim_run_stage.argtypes = [vp, i32, vp]
    lib.dexsim_time_stage.argtypes = [vp, i32, i32, vp, P(f32)]
    lib.dexsim_step_timing.argtypes = [vp, i32, P(f32), P(i32)]
    lib.dexsim_set_step_sink.argtypes = [vp, vp, vp, vp]
    lib.dexsim_set_action_copy.argtypes = [vp, vp]
    lib.dexsim_set_stats_sink.argtypes = [vp, vp]
    lib.dexsim_set_phase_probe.argtypes = [vp, vp]
    lib.dexsim_set_obs_dict_mode.argtypes = [vp, i32]
    for name in EXPORTED_SYMBOLS:
        getattr(lib, name).restype = i32
    lib.dexsim_error_string.argtypes = [i32]
    lib.dexsim_error_string.restype = C.c_char_p
    lib.dexsim_last_error.argtypes = []
    lib.dexsim_last_error.restype = C.c_char_p
    return lib


def check_struct_sizes(lib):
    out = (C.c_size_t * 4)()
    lib.dexsim_struct_sizes(out)
    mine = [C.sizeof(DexHandModel), C.sizeof(DexSimConfig), C.sizeof(DexSimField), C.sizeof(DexSimBuffers)]
    if list(out) != mine:
        raise RuntimeError(f"dexsim ABI mismatch: library struct sizes {list(out)} != python mirrors {mine}")
