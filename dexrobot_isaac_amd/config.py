"""Resolved configuration dicts for the two tasks, and their translation into the C-ABI DexSimConfig.

The reference feeds the env a plain resolved dict with sections sim / env / task / train plus a root
`physics_engine` (reference train.py:246-253).  Hydra/omegaconf are not available here, so the
defaults below restate the *values* of the reference's YAML tree (file:line cited per block) as Python
literals; `make_env(cfg=...)` accepts any dict with the same keys (e.g. one a user resolved with Hydra).
"""
import copy
import math

from . import _abi
from .hand_model import HandModel

# reference constants.py:57-88 FINGER_COUPLING_MAP: finger control -> [(dof name, scale)]
FINGER_COUPLING_MAP = {
    0: [("r_f_joint1_1", 1.0)],
    1: [("r_f_joint1_2", 1.0)],
    2: [("r_f_joint1_3", 1.0), ("r_f_joint1_4", 1.0)],
    3: [("r_f_joint2_1", 1.0), ("r_f_joint4_1", 1.0), ("r_f_joint5_1", 2.0)],
    4: [("r_f_joint2_2", 1.0)],
    5: [("r_f_joint2_3", 1.0), ("r_f_joint2_4", 1.0)],
    6: [("r_f_joint3_2", 1.0)],
    7: [("r_f_joint3_3", 1.0), ("r_f_joint3_4", 1.0)],
    8: [("r_f_joint4_2", 1.0)],
    9: [("r_f_joint4_3", 1.0), ("r_f_joint4_4", 1.0)],
    10: [("r_f_joint5_2", 1.0)],
    11: [("r_f_joint5_3", 1.0), ("r_f_joint5_4", 1.0)],
}

# reference hand_initializer.py:20-38 HardwareMapping (enum order = obs order of the 12 active controls)
HARDWARE_MAPPING = [
    ("th_dip", ["r_f_joint1_3", "r_f_joint1_4"]), ("th_mcp", ["r_f_joint1_2"]), ("th_rot", ["r_f_joint1_1"]),
    ("ff_spr", ["r_f_joint2_1", "r_f_joint4_1", "r_f_joint5_1"]), ("ff_dip", ["r_f_joint2_3", "r_f_joint2_4"]),
    ("ff_mcp", ["r_f_joint2_2"]), ("mf_dip", ["r_f_joint3_3", "r_f_joint3_4"]), ("mf_mcp", ["r_f_joint3_2"]),
    ("rf_dip", ["r_f_joint4_3", "r_f_joint4_4"]), ("rf_mcp", ["r_f_joint4_2"]),
    ("lf_dip", ["r_f_joint5_3", "r_f_joint5_4"]), ("lf_mcp", ["r_f_joint5_2"]),
]

_PHYSX_DEFAULT = {  # cfg/physics/default.yaml:3-20
    "solver_type": 1, "num_position_iterations": 16, "num_velocity_iterations": 0,
    "contact_offset": 0.001, "rest_offset": 0.0005, "bounce_threshold_velocity": 0.15,
    "max_depenetration_velocity": 0.2, "default_buffer_size_multiplier": 4.0, "num_subscenes": 0,
    "contact_collection": 1, "gpu_contact_pairs_per_env": 512, "always_use_articulations": True,
    "num_threads": 4,
}

_BASE_TASK = {
    "physics_engine": "physx",                                   # cfg/config.yaml:17
    "sim": {"dt": 0.005, "substeps": 4, "gravity": [0.0, 0.0, -9.81], "num_client_threads": 0,
            "physics_engine": "physx", "graphicsDeviceId": 0,
            "physx": dict(_PHYSX_DEFAULT)},                      # cfg/task/BaseTask.yaml:8-11
    "env": {"numEnvs": 1024, "device": "cuda:0", "viewer": False, "videoRecord": False, "videoStream": False,
            "controlMode": "position", "clipObservations": math.inf, "clipActions": math.inf,
            "envSpacing": 2.0, "initialHandPos": [0.0, 0.0, 0.5], "initialHandRot": [0.0, 0.0, 0.0, 1.0],
            "episodeLength": 300},                               # cfg/task/BaseTask.yaml:14-23, config.yaml:44-53
    "task": {
        "name": "BaseTask",
        "controlMode": "position_delta",                          # cfg/task/BaseTask.yaml:31
        "policy_controls_hand_base": True, "policy_controls_fingers": True,
        "default_base_targets": [0.0] * 6, "default_finger_targets": [0.0] * 12,
        "max_finger_joint_velocity": 1.0, "max_base_linear_velocity": 0.5,
        "max_base_angular_velocity": 1.5,                         # :43-45
        "termination": {"active_success_criteria": [], "active_failure_criteria": [],
                        "height_safety": {"handbase_threshold": 0.0, "fingertip_threshold": 0.0,
                                          "fingerpad_threshold": 0.0}},
        "reward_weights": {                                       # :60-76
            "alive": 0.1, "height_safety": 1.0, "finger_velocity": 0.1, "hand_velocity": 0.1,
            "hand_angular_velocity": 0.1, "joint_limit": 0.5, "finger_acceleration": 0.05,
            "hand_acceleration": 0.05, "hand_angular_acceleration": 0.05, "contact_stability": 0.2,
            "termination_success": 10.0, "termination_failure_penalty": 5.0,
            "termination_timeout_penalty": 0.0},
        "enable_component_debug_logs": False,
        "max_consecutive_successes": 50,
        "contact_force_bodies": [f"r_f_link{f}_4" for f in range(1, 6)],   # :84-89
        "contact_binary_threshold": 1.0,
        "policy_observation_keys": [                              # :103-116
            "base_dof_pos", "base_dof_vel", "active_finger_dof_pos", "active_finger_dof_vel", "hand_pose",
            "contact_binary", "prev_actions", "base_dof_target", "active_finger_dof_target",
            "fingertip_poses_world", "fingertip_poses_hand", "fingerpad_poses_world", "fingerpad_poses_hand"],
    },
    "train": {"seed": 42},                                        # cfg/config.yaml:60
}


def _blind_grasping():
    cfg = copy.deepcopy(_BASE_TASK)
    cfg["sim"]["dt"] = 0.01                                       # cfg/task/BlindGrasping.yaml:9
    cfg["env"].update({
        "episodeLength": 500,                                     # :15
        "box": {"size": 0.05, "mass": 0.1, "friction": 1.0, "restitution": 0.0,
                "initial_position": {"xy_range": 0.02, "z": 0.027}},   # :17-24
    })
    t = cfg["task"]
    t.update({
        "name": "BlindGrasping",
        "max_base_linear_velocity": 0.1, "max_base_angular_velocity": 0.5,
        "max_finger_joint_velocity": 1.0,                         # :36-38
        "contact_binary_threshold": 0.1,                          # :41
        "penetration_prevention": {"geometricPenetrationFactor": 1.0, "proximityMinDistanceFactor": 1.0,
                                   "penetrationDepthScale": 100.0},
        "policy_observation_keys": [                              # :50-66
            "base_dof_pos", "active_finger_dof_pos", "base_dof_vel", "active_finger_dof_vel",
            "contact_binary", "contact_duration", "hand_pose", "prev_actions", "fingertip_poses_world",
            "fingerpad_poses_world", "first_three_fingerpad_centroid", "episode_time", "fingerpad_distances",
            "current_stage", "time_in_stage", "stage_progress"],
        "termination": {
            "active_success_criteria": ["grasp_lift_success"],
            "active_failure_criteria": ["hitting_ground", "box_too_far", "stage1_pregrasp_failed",
                                        "stage2_contact_failed", "stage3_grasp_lost"],
            "height_safety": {"handbase_threshold": 0.0, "fingertip_threshold": 0.0,
                              "fingerpad_threshold": 0.0}},
        "success_height_threshold": 0.2, "contact_duration_threshold": 2.0, "min_fingers_for_grasp": 2,
        "max_box_distance": 0.8, "stage1_duration": 4.0, "stage2_duration": 3.0,
        "hand_translation_range": 0.2, "hand_rotation_range": 0.785,
        "finger_randomization": {"thumb_rotation_range": 1.571, "other_finger_range": 0.524},
        "stage_evaluation": {"stage2_contact_success_threshold": 0.5},
        "reward_calculation": {"height_alignment_decay": 5.0, "centroid_positioning_decay": 5.0,
                               "object_stability_decay": 100.0, "first_three_height_consistency_decay": 50.0,
                               "fingerpad_proximity_decay": 10.0, "base_stability_decay": 3.0},
        "quality_thresholds": {"height_tolerance": 0.03, "centroid_tolerance": 0.08,
                               "position_drift_tolerance": 0.01, "velocity_tolerance": 0.005},
        "visualization": {"box_color": [0.5, 0.8, 1.0]},
        "reward_weights": {                                       # :128-173 (parent weights deleted)
            "s1_height_alignment": 0.3, "s1_centroid_positioning": 1.0, "s1_object_stability": 0.2,
            "s1_finger_height_consistency": 0.4, "s1_thumb_rotation": 0.3,
            "s2_thumb_contact": 1.5, "s2_other_fingers_contact": 1.0, "s2_grasp_achievement": 5.0,
            "s2_fingerpad_proximity": 2.5, "s2_base_stability": 2.0,
            "s3_object_height": 10.0, "s3_grasp_maintenance": 10.0, "s3_grasp_duration": 2.0,
            "s1_completion": 500.0, "s2_completion": 1000.0, "penetration_penalty": -100.0,
            "termination_success": 2000.0, "termination_failure_penalty": 200.0,
            "termination_timeout_penalty": 200.0,
            "alive": 0.0, "height_safety": 0.0, "finger_velocity": 0.0, "hand_velocity": 0.0,
            "hand_angular_velocity": 0.0, "joint_limit": 0.0, "finger_acceleration": 0.0,
            "hand_acceleration": 0.0, "hand_angular_acceleration": 0.0, "contact_stability": 0.0},
    })
    return cfg


def default_cfg(task_name):
    """Resolved cfg dict for `task_name` (BaseTask | BlindGrasping); ValueError otherwise (factory.py:61-62)."""
    if task_name == "BaseTask":
        return copy.deepcopy(_BASE_TASK)
    if task_name == "BlindGrasping":
        return _blind_grasping()
    raise ValueError(f"Unknown task: {task_name}")


# --------------------------------------------------------------------------- key tables from the ABI header
OBS_KEYS = [
    ("base_dof_pos", 6), ("base_dof_vel", 6), ("active_finger_dof_pos", 12), ("active_finger_dof_vel", 12),
    ("all_finger_dof_pos", 20), ("all_finger_dof_vel", 20), ("hand_pose", 7), ("hand_pose_arr_aligned", 7),
    ("contact_forces", 15), ("prev_actions", 18), ("active_prev_targets", 18), ("base_dof_target", 6),
    ("active_finger_dof_target", 12), ("all_finger_dof_target", 20), ("contact_force_magnitude", 5),
    ("contact_binary", 5), ("contact_duration", 5), ("fingertip_poses_world", 35),
    ("fingertip_poses_hand", 35), ("fingerpad_poses_world", 35), ("fingerpad_poses_hand", 35),
    ("episode_time", 1), ("active_rule_targets", 18),
    ("object_pos", 3), ("object_vel", 3), ("finger_to_object_distances", 5),
    ("avg_finger_to_object_distance", 1), ("finger_to_object_height_diff", 5),
    ("avg_finger_to_object_height_diff", 1), ("hand_to_object_distance", 1), ("fingerpad_distances", 10),
    ("first_three_fingerpad_centroid", 3), ("thumb_contact", 1), ("other_fingers_contact", 1),
    ("grasp_state", 1), ("grasp_duration", 1), ("current_stage", 1), ("time_in_stage", 1),
    ("stage_progress", 1),
]
NUM_BASE_TASK_OBS_KEYS = 23  # keys that exist without a task (observation_encoder.py:576-758)
REWARD_TERMS = [
    "alive", "height_safety", "finger_velocity", "hand_velocity", "hand_angular_velocity", "joint_limit",
    "finger_acceleration", "hand_acceleration", "hand_angular_acceleration", "contact_stability",
    "s1_height_alignment", "s1_centroid_positioning", "s1_object_stability", "s1_finger_height_consistency",
    "s1_thumb_rotation", "s2_thumb_contact", "s2_other_fingers_contact", "s2_grasp_achievement",
    "s2_fingerpad_proximity", "s2_base_stability", "s3_object_height", "s3_grasp_maintenance",
    "s3_grasp_duration", "s1_completion", "s2_completion", "penetration_penalty",
]


def obs_key_offsets():
    off, out = 0, {}
    for name, dim in OBS_KEYS:
        out[name] = (off, dim)
        off += dim
    assert off == _abi.OBS_ALL_DIM
    return out


def active_limits(model):
    """18 active limits: base 6 + first joint of each coupling group (action_processor.py:380-413)."""
    lo = [float(model.lo[j]) for j in range(6)]
    hi = [float(model.hi[j]) for j in range(6)]
    for c in range(12):
        j = model.dof_names.index(FINGER_COUPLING_MAP[c][0][0])
        lo.append(float(model.lo[j]))
        hi.append(float(model.hi[j]))
    return lo, hi


def build_sim_config(cfg, model=None, dr=None):
    """Translate a resolved cfg dict into the C-ABI DexSimConfig.

    Reproduces the derived quantities the reference computes at init: control_dt = 2*sim.dt
    (dexhand_base.py:270-320), max_deltas (action_processor.py:416-434), num_actions
    (initialization_manager.py:51-61), obs dimension (observation_encoder.py:207-234),
    contact_duration_threshold_steps = int(t/control_dt) (blind_grasping_task.py:271-273).
    """
    if "use_gpu_pipeline" in cfg.get("sim", {}) and not cfg["sim"].get("_dexsim_internal", False):
        # vec_task.py:66-71: deprecated key must be absent
        raise RuntimeError("The 'use_gpu_pipeline' config key is deprecated and must be removed.")
    if cfg["physics_engine"] != "physx":
        raise ValueError(f"Invalid physics engine backend: {cfg['physics_engine']}")   # vec_task.py:188-190
    sim, env, task = cfg["sim"], cfg["env"], cfg["task"]
    if model is None:
        model = HandModel(env.get("initialHandPos", (0, 0, 0.5)), env.get("initialHandRot", (0, 0, 0, 1)))
    c = _abi.DexSimConfig()
    c.num_envs = int(env["numEnvs"])
    name = task["name"]
    if name not in ("BaseTask", "BlindGrasping"):
        raise ValueError(f"Unknown task: {name}")
    c.task = _abi.TASK_BLIND_GRASPING if name == "BlindGrasping" else _abi.TASK_BASE
    c.dt = float(sim["dt"])
    c.substeps = int(sim["substeps"])
    for i in range(3):
        c.gravity[i] = float(sim["gravity"][i])
    px = sim["physx"]
    c.num_position_iterations = int(px["num_position_iterations"])
    c.contact_offset = float(px["contact_offset"])
    c.rest_offset = float(px["rest_offset"])
    c.max_depenetration_velocity = float(px["max_depenetration_velocity"])
    c.erp = float(sim.get("dexsim_erp", 0.2))
    # sim.dexsim_joint_limit_rows (new key, default off): joint limits as unilateral rows of the contact solver for finger joints
    # within sim.dexsim_joint_limit_margin (rad) of a limit, in envs with hand contacts (include/dexsim.h: joint_limit_rows)
    c.joint_limit_rows = int(bool(sim.get("dexsim_joint_limit_rows", False)))
    c.joint_limit_margin = float(sim.get("dexsim_joint_limit_margin", 0.01))
    control_dt = 2.0 * float(sim["dt"])   # python double, as PhysicsManager computes it (physics_manager.py:259)
    c.control_dt = control_dt
    c.episode_length = int(env["episodeLength"])
    c.seed = int(cfg["train"]["seed"]) & 0xFFFFFFFF

    mode = task["controlMode"]
    if mode not in ("position", "position_delta"):
        raise ValueError(f"Invalid control mode: {mode}")                      # action_processor.py:198-199
    c.control_mode = _abi.MODE_POSITION if mode == "position" else _abi.MODE_POSITION_DELTA
    c.policy_controls_base = int(bool(task["policy_controls_hand_base"]))
    c.policy_controls_fingers = int(bool(task["policy_controls_fingers"]))
    c.num_actions = 6 * c.policy_controls_base + 12 * c.policy_controls_fingers
    if c.num_actions == 0:
        raise RuntimeError("policy controls neither the hand base nor the fingers: empty action space")
    for i in range(18):
        lim = (task["max_base_linear_velocity"] if i < 3 else
               task["max_base_angular_velocity"] if i < 6 else task["max_finger_joint_velocity"])
        c.max_deltas[i] = control_dt * float(lim)
    lo, hi = active_limits(model)
    for i in range(18):
        c.active_lower[i], c.active_upper[i] = lo[i], hi[i]

    c.contact_binary_threshold = float(task["contact_binary_threshold"])
    offs = obs_key_offsets()
    valid = [k for k, _ in OBS_KEYS[:NUM_BASE_TASK_OBS_KEYS]] if c.task == _abi.TASK_BASE else list(offs)
    keys = task["policy_observation_keys"]
    if len(keys) > _abi.MAX_OBS_SEG:
        raise RuntimeError("too many policy_observation_keys")
    n = 0
    for i, k in enumerate(keys):
        if k not in valid:
            raise RuntimeError(f"Observation key '{k}' MISSING during initialization - fail fast")
        off, dim = offs[k]
        if k == "prev_actions":
            dim = c.num_actions
        c.obs_seg_off[i], c.obs_seg_len[i] = off, dim
        n += dim
    c.n_obs_seg, c.num_obs = len(keys), n

    term = task["termination"]
    hs = term.get("height_safety")
    c.height_safety_enabled = int(hs is not None)
    if hs is not None:
        c.handbase_threshold = float(hs["handbase_threshold"])
        c.fingertip_threshold = float(hs["fingertip_threshold"])
    succ_avail = _abi.SUCCESS_CRITERIA if c.task == _abi.TASK_BLIND_GRASPING else []
    fail_avail = (["hitting_ground"] if hs is not None else []) + (
        _abi.FAILURE_CRITERIA[1:] if c.task == _abi.TASK_BLIND_GRASPING else [])

    def mask(active, avail, names, kind):
        # termination_manager.py:98-118: an active criterion that nobody implements is fatal
        for a in active:
            if a not in avail:
                raise RuntimeError(f"{kind} criterion '{a}' is configured as active but not implemented! "
                                   f"Available criteria: {sorted(avail)}.")
        use = active if active else avail     # empty list = all available
        return sum(1 << names.index(a) for a in use)

    c.active_success_mask = mask(term.get("active_success_criteria", []), succ_avail,
                                 _abi.SUCCESS_CRITERIA, "Success")
    c.active_failure_mask = mask(term.get("active_failure_criteria", []), fail_avail,
                                 _abi.FAILURE_CRITERIA, "Failure")
    rw = task["reward_weights"]
    c.success_reward = float(rw["termination_success"])
    c.failure_penalty = float(rw["termination_failure_penalty"])
    c.timeout_penalty = float(rw["termination_timeout_penalty"])
    c.max_consecutive_successes = int(task["max_consecutive_successes"])
    for i, nme in enumerate(REWARD_TERMS):
        w = float(rw.get(nme, 0.0))
        if c.task == _abi.TASK_BASE and i >= _abi.NUM_COMMON_REWARD_TERMS:
            w = 0.0
        c.reward_weight[i] = w

    c.ground_friction = 0.5                                                    # dexhand_base.py:635-636
    # env.box.fixed (new key): the harness's static contact-test box (examples/dexhand_test.py:950-1024, gym.create_box with
    # fix_base_link = True, 0.1 m cube centred at (0.3, 0, 0.05)) as a config option -- for either task
    fixed_box = bool((env.get("box") or {}).get("fixed", False))
    c.has_box = int(c.task == _abi.TASK_BLIND_GRASPING or fixed_box)
    if fixed_box:
        box = env["box"]
        c.box_fixed = 1
        c.box_size, c.box_mass, c.box_friction = float(box.get("size", 0.1)), 1.0, float(box.get("friction", 1.0))
        pos = box.get("position", (0.3, 0.0, 0.5 * c.box_size))
        for i in range(3):
            c.box_fixed_pos[i] = float(pos[i])
        c.box_z = float(pos[2])
    if c.task == _abi.TASK_BLIND_GRASPING:
        box = env["box"]
        if not fixed_box:
            c.box_size, c.box_mass, c.box_friction = float(box["size"]), float(box["mass"]), float(box["friction"])
            c.box_z = float(box["initial_position"]["z"])
        c.box_xy_range = float(box["initial_position"]["xy_range"])
        c.height_threshold = float(task["success_height_threshold"])
        c.contact_duration_threshold_s = float(task["contact_duration_threshold"])
        c.contact_duration_threshold_steps = int(float(task["contact_duration_threshold"]) / control_dt)
        c.min_fingers_for_grasp = int(task["min_fingers_for_grasp"])
        c.max_box_distance = float(task["max_box_distance"])
        c.stage1_duration, c.stage2_duration = float(task["stage1_duration"]), float(task["stage2_duration"])
        c.hand_translation_range = float(task["hand_translation_range"])
        c.hand_rotation_range = float(task["hand_rotation_range"])
        fr = task["finger_randomization"]
        c.thumb_rotation_range, c.other_finger_range = float(fr["thumb_rotation_range"]), float(fr["other_finger_range"])
        c.stage2_contact_success_threshold = float(task["stage_evaluation"]["stage2_contact_success_threshold"])
        rc = task["reward_calculation"]
        c.height_alignment_decay = float(rc["height_alignment_decay"])
        c.centroid_positioning_decay = float(rc["centroid_positioning_decay"])
        c.object_stability_decay = float(rc["object_stability_decay"])
        c.first_three_height_consistency_decay = float(rc["first_three_height_consistency_decay"])
        c.fingerpad_proximity_decay = float(rc["fingerpad_proximity_decay"])
        c.base_stability_decay = float(rc["base_stability_decay"])
        pp = task["penetration_prevention"]
        c.geometric_penetration_factor = float(pp["geometricPenetrationFactor"])
        c.proximity_min_distance_factor = float(pp["proximityMinDistanceFactor"])
        c.penetration_depth_scale = float(pp["penetrationDepthScale"])
        qt = task["quality_thresholds"]
        c.height_tolerance, c.centroid_tolerance = float(qt["height_tolerance"]), float(qt["centroid_tolerance"])
        c.position_drift_tolerance = float(qt["position_drift_tolerance"])
        c.velocity_tolerance = float(qt["velocity_tolerance"])
    if dr:
        c.dr_enabled = 1
        c.dr_mass_lo, c.dr_mass_hi = [float(x) for x in dr["mass"]]
        c.dr_mu_lo, c.dr_mu_hi = [float(x) for x in dr["friction"]]
        c.dr_seed = int(dr.get("seed", 4242)) & 0xFFFFFFFF
    return c, model
