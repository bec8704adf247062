#!/usr/bin/env python
"""Generate golden vectors for the post-physics (L2) path by RUNNING THE REFERENCE'S OWN PYTHON CLASSES.

Run in the build container only (needs /root/reference; skips when absent).  Nothing from the reference is
copied: its modules are imported by path and driven through a fake parent env that carries synthetic L1
(sim-state) tensors.  The committed artefacts are DATA: inputs + the outputs the reference produced.

Third-party modules the reference imports but this image lacks are replaced by build-owned stand-ins that
contain no physics or task logic:
  loguru.logger            -> no-op logger
  gym.spaces.Box           -> shape holder
  isaacgym.gymapi/gymtorch -> empty shims (never called with effect)
  isaacgym.torch_utils     -> quat_mul / quat_conjugate / quat_rotate / quat_rotate_inverse /
                              quat_from_euler_xyz restated from the published xyzw-Hamilton definitions.
The quaternion stand-ins are PINNED by running the reference's own known-answer script
dexhand_env/utils/test_coordinate_transforms.py against them (it must print all 7 cases passed);
everything else exercised below (ActionProcessor/ActionRules/DefaultActionRules/ActionScaling,
ObservationEncoder, BlindGraspingTask, StepProcessor, TerminationManager, RewardCalculator, ResetManager,
InitializationManager) is reference code executing unmodified.

The orchestration the fake parent performs (step / reset / reset_idx call order) restates
dexhand_env/tasks/dexhand_base.py:743-942, which cannot be imported (it pulls in the viewer/video stack).

Output: tests/golden/l2_<scenario>.npz  (+ the cfg overrides as JSON inside the npz).
"""
import copy
import json
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)


# ----------------------------------------------------------------------------------------------- stand-ins
def install_standins():
    class _Logger:
        def __getattr__(self, name):
            return lambda *a, **k: None

    loguru = types.ModuleType("loguru")
    loguru.logger = _Logger()
    sys.modules["loguru"] = loguru

    gym = types.ModuleType("gym")
    spaces = types.ModuleType("gym.spaces")

    class Box:
        def __init__(self, low=None, high=None, shape=None, dtype=None):
            self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

    spaces.Box = Box
    gym.spaces = spaces
    sys.modules["gym"] = gym
    sys.modules["gym.spaces"] = spaces

    isaacgym = types.ModuleType("isaacgym")
    gymapi = types.ModuleType("isaacgym.gymapi")
    gymapi.DOF_MODE_POS = 1
    gymapi.DOMAIN_SIM = 0

    class _Any:
        def __init__(self, *a, **k):
            pass

    for n in ("AssetOptions", "Transform", "Vec3", "Quat", "SimParams", "PlaneParams"):
        setattr(gymapi, n, _Any)
    gymtorch = types.ModuleType("isaacgym.gymtorch")
    gymtorch.unwrap_tensor = lambda t: t
    gymtorch.wrap_tensor = lambda t: t
    tu = types.ModuleType("isaacgym.torch_utils")

    def quat_mul(a, b):
        x1, y1, z1, w1 = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
        x2, y2, z2, w2 = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
        return torch.stack([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                            w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                            w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2,
                            w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2], dim=-1)

    def quat_conjugate(a):
        return torch.cat([-a[..., :3], a[..., 3:]], dim=-1)

    def quat_rotate(q, v):
        qw, qv = q[:, 3:4], q[:, :3]
        return v * (2.0 * qw ** 2 - 1.0) + torch.cross(qv, v, dim=-1) * qw * 2.0 + \
            qv * (qv * v).sum(-1, keepdim=True) * 2.0

    def quat_rotate_inverse(q, v):
        qw, qv = q[:, 3:4], q[:, :3]
        return v * (2.0 * qw ** 2 - 1.0) - torch.cross(qv, v, dim=-1) * qw * 2.0 + \
            qv * (qv * v).sum(-1, keepdim=True) * 2.0

    def quat_from_euler_xyz(roll, pitch, yaw):
        cy, sy = torch.cos(yaw * 0.5), torch.sin(yaw * 0.5)
        cr, sr = torch.cos(roll * 0.5), torch.sin(roll * 0.5)
        cp, sp = torch.cos(pitch * 0.5), torch.sin(pitch * 0.5)
        return torch.stack([cy * sr * cp - sy * cr * sp, cy * cr * sp + sy * sr * cp,
                            sy * cr * cp - cy * sr * sp, cy * cr * cp + sy * sr * sp], dim=-1)

    tu.quat_mul, tu.quat_conjugate, tu.quat_rotate = quat_mul, quat_conjugate, quat_rotate
    tu.quat_rotate_inverse, tu.quat_from_euler_xyz = quat_rotate_inverse, quat_from_euler_xyz
    isaacgym.gymapi, isaacgym.gymtorch, isaacgym.torch_utils = gymapi, gymtorch, tu
    sys.modules.update({"isaacgym": isaacgym, "isaacgym.gymapi": gymapi, "isaacgym.gymtorch": gymtorch,
                        "isaacgym.torch_utils": tu})


# ----------------------------------------------------------------------------------------------- fake parent
class FakeGym:
    def __getattr__(self, name):
        return lambda *a, **k: True


class FakeTensorManager:
    """Shapes/semantics of components/physics/tensor_manager.py:139-459 (views + per-step gather copy)."""

    def __init__(self, parent, dof_props):
        self.parent = parent
        self.device = parent.device
        self.dof_props = dof_props
        self.tensors_initialized = True

    def install(self, dof_state, root, rbs, cf_all, idx5):
        self.dof_state = dof_state
        self.dof_pos = dof_state[..., 0]
        self.dof_vel = dof_state[..., 1]
        self.actor_root_state_tensor = root
        self.rigid_body_states = rbs
        self.contact_forces_all = cf_all
        self.contact_forces = cf_all[:, torch.tensor(idx5, dtype=torch.long), :]

    def refresh_tensors(self, idx5):
        self.contact_forces = self.contact_forces_all[:, torch.tensor(idx5, dtype=torch.long), :]


class FakePhysicsManager:
    def __init__(self, parent, physics_dt):
        self.parent = parent
        self.physics_dt = physics_dt
        self.control_dt = None
        self.physics_steps_per_control_step = 2
        self.n_steps = 0

    def step_physics(self, refresh_tensors=True):
        self.n_steps += 1
        self.parent.physics_callback()
        return True

    def apply_dof_states(self, env_ids, dof_state, actor_index=0):
        return True

    def mark_control_step(self):
        pass


class FakeParent:
    def __init__(self, cfg, model, task_name, l1_script):
        from dexhand_env.components.action.action_processor import ActionProcessor
        from dexhand_env.components.action.default_rules import DefaultActionRules
        from dexhand_env.components.initialization.hand_initializer import HandInitializer
        from dexhand_env.components.initialization.initialization_manager import InitializationManager
        from dexhand_env.components.observation.observation_encoder import ObservationEncoder
        from dexhand_env.components.reset.reset_manager import ResetManager
        from dexhand_env.components.reward.reward_calculator import RewardCalculator
        from dexhand_env.components.step_processor import StepProcessor
        from dexhand_env.components.termination.termination_manager import TerminationManager
        from dexhand_env.constants import BASE_JOINT_NAMES, FINGERPAD_BODY_NAMES, FINGERTIP_BODY_NAMES
        from dexhand_env.tasks.base_task import BaseTask
        from dexhand_env.tasks.blind_grasping_task import BlindGraspingTask

        self.cfg, self.env_cfg, self.task_cfg, self.sim_cfg = cfg, cfg["env"], cfg["task"], cfg["sim"]
        self.num_envs = cfg["env"]["numEnvs"]
        self.device = "cpu"
        self.gym, self.sim = FakeGym(), None
        self.model = model
        self.l1 = l1_script
        self.l1_cursor = 0
        N, B = self.num_envs, len(model.body_names) + (1 if task_name == "BlindGrasping" else 0)
        A = 2 if task_name == "BlindGrasping" else 1
        self.base_joint_names = BASE_JOINT_NAMES
        self.fingertip_body_names, self.fingerpad_body_names = FINGERTIP_BODY_NAMES, FINGERPAD_BODY_NAMES
        # live "engine" tensors
        self.dof_state = torch.zeros(N, 26, 2)
        self.actor_root_state_tensor = torch.zeros(N, A, 13)
        self.rigid_body_states = torch.zeros(N, B, 13)
        self.contact_forces_all = torch.zeros(N, B, 3)
        self.dof_pos, self.dof_vel = self.dof_state[..., 0], self.dof_state[..., 1]

        if task_name == "BlindGrasping":
            self.task = BlindGraspingTask(None, None, torch.device("cpu"), N, cfg)
        else:
            self.task = BaseTask(None, None, torch.device("cpu"), N, cfg)
        self.task.parent_env = self
        self.hand_initializer = HandInitializer(parent=self, asset_root="/nonexistent")
        self.hand_initializer._dof_names = list(model.dof_names)
        self.hand_initializer.contact_force_body_names = cfg["task"]["contact_force_bodies"]
        self.hand_initializer.fingertip_local_indices = model.fingertip_local_indices
        self.hand_initializer.fingerpad_local_indices = model.fingerpad_local_indices
        self.hand_local_rigid_body_index = model.hand_local_rigid_body_index
        self.hand_local_actor_index = 0
        self.contact_force_local_body_indices = model.body_indices(cfg["task"]["contact_force_bodies"])
        self.fingertip_local_indices = model.fingertip_local_indices
        self.fingerpad_local_indices = model.fingerpad_local_indices
        self.tensor_manager = FakeTensorManager(self, torch.tensor(model.dof_props()))
        self.tensor_manager.install(self.dof_state, self.actor_root_state_tensor, self.rigid_body_states,
                                    self.contact_forces_all, self.contact_force_local_body_indices)
        self.contact_forces = self.tensor_manager.contact_forces
        self.dof_props = self.tensor_manager.dof_props
        if task_name == "BlindGrasping":
            # what set_tensor_references (blind_grasping_task.py:368-431) resolves through Isaac Gym lookups
            t = self.task
            t.root_state_tensor = self.actor_root_state_tensor
            t.box_actor_indices = torch.arange(N) * 2 + 1
            t.box_local_actor_index = 1
            t.box_local_rigid_body_index = B - 1
            t.box_states = self.actor_root_state_tensor[:, 1, :]
            t.box_positions = t.box_states[:, :3]
            t.box_velocities = t.box_states[:, 7:10]
            t.initial_box_positions = torch.zeros((N, 3))
            t.initial_box_positions[:, 2] = t.box_z
        self.physics_manager = FakePhysicsManager(self, cfg["sim"]["dt"])
        self.action_processor = ActionProcessor(parent=self)
        self.num_dof = 26
        self.action_processor.initialize_from_config({
            "control_mode": cfg["task"]["controlMode"], "num_dof": 26,
            "policy_controls_hand_base": cfg["task"]["policy_controls_hand_base"],
            "policy_controls_fingers": cfg["task"]["policy_controls_fingers"],
            "finger_vel_limit": cfg["task"]["max_finger_joint_velocity"],
            "base_lin_vel_limit": cfg["task"]["max_base_linear_velocity"],
            "base_ang_vel_limit": cfg["task"]["max_base_angular_velocity"],
            "post_action_filters": ["velocity_clamp", "position_clamp"]})
        self.observation_encoder = ObservationEncoder(parent=self)
        self.reset_manager = ResetManager(parent=self, dof_state=self.dof_state,
                                          root_state_tensor=self.actor_root_state_tensor,
                                          hand_local_actor_index=0, default_dof_pos=torch.zeros(26),
                                          task=self.task)
        self.termination_manager = TerminationManager(parent=self, task_cfg=self.task_cfg)
        self.reward_calculator = RewardCalculator(parent=self, task_cfg=self.task_cfg)
        self.initialization_manager = InitializationManager(parent=self)
        self.step_processor = StepProcessor(parent=self)
        self.obs_dict = {}
        self.initialization_manager.create_index_mappings()
        self.initialization_manager.setup_additional_tensors()
        DefaultActionRules.setup_default_action_rule(self.action_processor, cfg["task"]["controlMode"])
        # control-cycle measurement outcome (dexhand_base.py:270-320): control_dt = 2 * physics_dt
        self.physics_manager.control_dt = self.physics_manager.physics_dt * 2
        self.action_processor.finalize_setup()
        self.task.finalize_setup() if hasattr(self.task, "finalize_setup") else None
        self.rand_log = []

    # --- VecTask num_observations / num_actions are plain attributes here
    @property
    def episode_time(self):   # dexhand_base.py:702-712
        if self.physics_manager.control_dt is None:
            return torch.zeros(self.num_envs)
        return self.episode_step_count.float() * self.physics_manager.control_dt

    def physics_callback(self):
        """'gym.simulate': overwrite the engine tensors with the next scripted L1 state."""
        s = self.l1[self.l1_cursor]
        self.l1_cursor += 1
        if getattr(self, "l1_hook", None) is not None:
            self.l1_hook(self, s)            # may adapt the scripted state in place (recorded as modified)
        self.dof_state[..., 0] = torch.tensor(s["q"])
        self.dof_state[..., 1] = torch.tensor(s["qd"])
        self.rigid_body_states[:] = torch.tensor(s["rbs"])
        self.contact_forces_all[:] = torch.tensor(s["cf"])
        if self.actor_root_state_tensor.shape[1] > 1:
            self.actor_root_state_tensor[:, 1, :] = torch.tensor(s["box"])

    # --- dexhand_base.py:743-803 (minus video bookkeeping)
    def reset_idx(self, env_ids):
        if len(env_ids) == 0:
            return
        # for the per-step snapshots: the obs_dict the step's obs_buf was built from.  Several obs_dict entries are VIEWS of
        # the engine tensors (dof_pos, root / rigid-body state), so after the reset's physics step they show the post-reset
        # state of every env while obs_buf keeps the terminal observation.
        self.obs_dict_pre_reset = {k: v.clone() for k, v in self.obs_dict.items()} if self.obs_dict else None
        self.termination_manager.reset_tracking(env_ids)
        self.reset_manager.reset_idx(env_ids)
        self.observation_encoder.reset_observer_state(env_ids)

    # --- dexhand_base.py:805-838
    def reset(self):
        self.reset_idx(torch.arange(self.num_envs))
        obs_dict = self.observation_encoder.compute_observations(exclude_components=["active_rule_targets"])
        rule = self.action_processor.apply_pre_action_rule(self.action_processor.active_prev_targets,
                                                           {"obs_dict": obs_dict, "env": self})
        obs_dict["active_rule_targets"] = rule
        self.obs_buf = self.observation_encoder.concatenate_observations(obs_dict)
        self.obs_dict = obs_dict
        obs, _, _, _ = self.step_processor.process_physics_step()
        return obs

    # --- dexhand_base.py:840-942
    def step(self, actions):
        self.actions = actions.clone()
        self.action_processor.process_actions(actions=self.actions,
                                              active_rule_targets=self.obs_dict["active_rule_targets"])
        self.observation_encoder.update_prev_actions(self.actions)
        self.physics_manager.step_physics(refresh_tensors=True)
        return self.step_processor.process_physics_step()


# ----------------------------------------------------------------------------------------------- L1 script
def quat_norm(q):
    return q / np.linalg.norm(q, axis=-1, keepdims=True)


def make_l1_script(rng, model, N, T, has_box, profile, extra_states=0):
    """Synthetic sim-state sequence: 2 states per control step (main physics step, optional reset step)."""
    B = len(model.body_names) + (1 if has_box else 0)
    hb, tips, pads = model.hand_local_rigid_body_index, model.fingertip_local_indices, model.fingerpad_local_indices
    c5 = model.body_indices([f"r_f_link{f}_4" for f in range(1, 6)])
    lo, hi = model.lo, model.hi
    q = rng.uniform(0.05, 0.4, (N, 26)).astype(np.float32)
    q[:, :6] = rng.uniform(-0.1, 0.1, (N, 6))
    script = []
    box0 = np.zeros((N, 13), dtype=np.float32)
    box0[:, 0:2] = rng.uniform(-0.02, 0.02, (N, 2))
    box0[:, 2] = 0.0255
    box0[:, 6] = 1.0
    group = np.arange(N) % 6
    for t in range(2 * T + 4 + extra_states):
        k = t // 2
        q = np.clip(q + rng.normal(0, 0.01, q.shape).astype(np.float32), lo - 0.01, hi + 0.01).astype(np.float32)
        qd = rng.normal(0, 0.5, (N, 26)).astype(np.float32)
        rbs = rng.normal(0, 0.2, (N, B, 13)).astype(np.float32)
        rbs[:, :, 3:7] = quat_norm(rng.normal(0, 1, (N, B, 4))).astype(np.float32)
        hand = np.array([0.0, 0.0, 0.35], dtype=np.float32) + rng.normal(0, 0.02, (N, 3)).astype(np.float32)
        rbs[:, hb, :3] = hand
        rbs[:, hb, 7:13] = rng.normal(0, 0.3, (N, 6))
        box = box0.copy()
        box[:, 7:10] = rng.normal(0, 0.001, (N, 3))
        cf = np.zeros((N, B, 3), dtype=np.float32)
        for f in range(5):
            rbs[:, tips[f], :3] = hand + np.array([0.02 * (f - 2), 0.03, -0.15], dtype=np.float32) + \
                rng.normal(0, 0.01, (N, 3))
            rbs[:, pads[f], :3] = rbs[:, tips[f], :3] + rng.normal(0, 0.004, (N, 3))
        if has_box:
            cf[:, B - 1, 2] = 0.981
        if profile == "fast":
            # group 1: good pregrasp (pads level with the box and centred), then contact, lift -> success path
            g1 = group == 1
            for f in range(5):
                off = np.array([0.02 * (f - 2), 0.0, 0.0], dtype=np.float32)
                rbs[g1, pads[f], :3] = box0[g1, :3] + off + rng.normal(0, 0.002, (g1.sum(), 3))
                rbs[g1, tips[f], :3] = rbs[g1, pads[f], :3] + np.array([0, 0, 0.03], dtype=np.float32)
            box[g1, 7:10] = rng.normal(0, 0.0005, (g1.sum(), 3))
            if k >= 6:   # contact on thumb + index + middle, box loaded
                for f in (0, 1, 2):
                    cf[np.ix_(g1, [c5[f]])] = np.array([0.5, 0.2, 0.3], dtype=np.float32)
                cf[g1, B - 1, :] = np.array([0.6, 0.1, 1.2], dtype=np.float32)
            if k >= 12:  # lifted
                box[g1, 2] = 0.25
                for f in range(5):
                    off = np.array([0.02 * (f - 2), 0.0, 0.0], dtype=np.float32)
                    rbs[g1, pads[f], :3] = box[g1, :3] + off
                    rbs[g1, tips[f], :3] = rbs[g1, pads[f], :3] + np.array([0, 0, 0.03], dtype=np.float32)
            # group 2: policy-observable contact but no box force -> stage2_contact_failed / grasp lost
            g2 = group == 2
            for f in range(3):
                rbs[g2, pads[f], :3] = box0[g2, :3] + np.array([0.02 * (f - 1), 0, 0], dtype=np.float32)
                rbs[g2, tips[f], :3] = rbs[g2, pads[f], :3] + np.array([0, 0, 0.03], dtype=np.float32)
            if k >= 6:
                for f in (0, 1):
                    cf[np.ix_(g2, [c5[f]])] = np.array([0.0, 0.4, 0.0], dtype=np.float32)
                cf[g2, B - 1, :] = 0.0
            # group 3: fingertip goes below ground at k == 9; group 4: hand far from box from k == 14
            g3 = group == 3
            if k == 9:
                rbs[g3, tips[2], 2] = -0.004
            g4 = group == 4
            if k >= 14:
                rbs[g4, hb, 0] = 1.2
            # group 5: intermittent contacts (duration tracking) ; group 0: nominal
            g5 = group == 5
            if (k // 3) % 2 == 0:
                cf[np.ix_(g5, [c5[3]])] = np.array([0.3, 0.0, 0.2], dtype=np.float32)
        elif profile == "long":
            # default-length FSM (stage 1 = 4 s = 200 control steps, stage 2 <= 3 s, success after 2 s of lifted grasp):
            # group 1 keeps a good pre-grasp pose, touches from script state 205 (stage 3 by contact duration at ~230), lifts at
            # state 232 and holds -> grasp_lift_success; group 2 has policy-observable contact without box force ->
            # stage2_contact_failed; the other groups fail the pre-grasp check at control step ~199
            g1 = group == 1
            for f in range(5):
                off = np.array([0.02 * (f - 2), 0.0, 0.0], dtype=np.float32)
                rbs[g1, pads[f], :3] = box0[g1, :3] + off + rng.normal(0, 0.002, (g1.sum(), 3))
                rbs[g1, tips[f], :3] = rbs[g1, pads[f], :3] + np.array([0, 0, 0.03], dtype=np.float32)
            box[g1, 7:10] = rng.normal(0, 0.0005, (g1.sum(), 3))
            if t >= 205:
                for f in (0, 1, 2):
                    cf[np.ix_(g1, [c5[f]])] = np.array([0.5, 0.2, 0.3], dtype=np.float32)
                cf[g1, B - 1, :] = np.array([0.6, 0.1, 1.2], dtype=np.float32)
            if t >= 232:
                box[g1, 2] = 0.25
                for f in range(5):
                    off = np.array([0.02 * (f - 2), 0.0, 0.0], dtype=np.float32)
                    rbs[g1, pads[f], :3] = box[g1, :3] + off
                    rbs[g1, tips[f], :3] = rbs[g1, pads[f], :3] + np.array([0, 0, 0.03], dtype=np.float32)
            g2 = group == 2
            for f in range(5):
                off = np.array([0.02 * (f - 2), 0.0, 0.0], dtype=np.float32)
                rbs[g2, pads[f], :3] = box0[g2, :3] + off + rng.normal(0, 0.002, (g2.sum(), 3))
                rbs[g2, tips[f], :3] = rbs[g2, pads[f], :3] + np.array([0, 0, 0.03], dtype=np.float32)
            box[g2, 7:10] = rng.normal(0, 0.0005, (g2.sum(), 3))
            if t >= 205:
                for f in (0, 1):
                    cf[np.ix_(g2, [c5[f]])] = np.array([0.0, 0.4, 0.0], dtype=np.float32)
                cf[g2, B - 1, :] = 0.0
        else:
            if k % 7 in (2, 3, 4):
                cf[::2, c5[1], :] = rng.normal(0, 1.0, (len(range(0, N, 2)), 3))
            if k % 5 == 0:
                cf[1::3, c5[4], :] = rng.normal(0, 2.0, (len(range(1, N, 3)), 3))
        if has_box:
            rbs[:, B - 1, :] = box
        script.append({"q": q.copy(), "qd": qd, "rbs": rbs, "cf": cf, "box": box})
    return script


def pack_l1(s, model, has_box):
    """Compact L1 view the oracle / HIP kernels consume (arena fields, SoA [rows][N])."""
    hb, tips, pads = model.hand_local_rigid_body_index, model.fingertip_local_indices, model.fingerpad_local_indices
    N = s["q"].shape[0]
    site = np.zeros((11, 7, N), dtype=np.float32)
    for i, b in enumerate([hb] + tips + pads):
        site[i] = s["rbs"][:, b, :7].T
    cforce = np.zeros((17, 3, N), dtype=np.float32)
    for f in range(5):
        cforce[3 * f + 2] = s["cf"][:, model.body_names.index(f"r_f_link{f + 1}_4"), :].T
    out = {"q": s["q"].T.copy(), "qd": s["qd"].T.copy(), "site_pose": site.reshape(77, N),
           "hand_vel": s["rbs"][:, hb, 7:13].T.copy()}
    if has_box:
        cforce[16] = s["cf"][:, -1, :].T
        out.update({"box_pos": s["box"][:, 0:3].T.copy(), "box_quat": s["box"][:, 3:7].T.copy(),
                    "box_lin": s["box"][:, 7:10].T.copy(), "box_ang": s["box"][:, 10:13].T.copy()})
    out["cforce"] = cforce.reshape(51, N)
    return out


# ----------------------------------------------------------------------------------------------- scenarios
def scenario_cfg(name):
    from dexrobot_isaac_amd.config import default_cfg
    if name == "blind_default":
        cfg, over = default_cfg("BlindGrasping"), {}
    elif name == "blind_long":
        cfg, over = default_cfg("BlindGrasping"), {}
    elif name in ("blind_fast", "blind_wide"):
        cfg = default_cfg("BlindGrasping")
        over = {"env.episodeLength": 24, "task.stage1_duration": 0.1, "task.stage2_duration": 0.2,
                "task.stage_evaluation.stage2_contact_success_threshold": 0.06,
                "task.contact_duration_threshold": 0.1}
    elif name == "base_default":
        cfg, over = default_cfg("BaseTask"), {"env.episodeLength": 20}
    elif name in ("base_position", "base_wide"):
        cfg, over = default_cfg("BaseTask"), {"task.controlMode": "position", "env.episodeLength": 50 if name == "base_position" else 9}
    else:
        raise KeyError(name)
    for k, v in over.items():
        d = cfg
        parts = k.split(".")
        for p in parts[:-1]:
            d = d[p]
        d[parts[-1]] = v
    return cfg, over


def run_scenario(name, N, T, seed, events=None, snap_steps=()):
    """events: {step: [env ids]} -- an explicit env.reset_idx(ids) call (dexhand_base.py:743-803) BEFORE that step;
    snap_steps: steps after which every obs_dict key and every reward component is recorded."""
    from dexrobot_isaac_amd.config import OBS_KEYS
    from dexrobot_isaac_amd.hand_model import HandModel
    events = events or {}
    cfg, over = scenario_cfg(name)
    cfg["env"]["numEnvs"] = N
    task_name = cfg["task"]["name"]
    has_box = task_name == "BlindGrasping"
    model = HandModel()
    rng = np.random.default_rng(seed)
    profile = {"blind_fast": "fast", "blind_wide": "fast", "blind_long": "long"}.get(name, "default")
    script = make_l1_script(rng, model, N, T, has_box, profile, extra_states=len(events))
    torch.manual_seed(cfg["train"]["seed"])
    env = FakeParent(copy.deepcopy(cfg), model, task_name, script)

    # record every torch.rand the reference draws (reset randomisation), in call order
    real_rand = torch.rand
    draws = []

    def rec_rand(*a, **k):
        out = real_rand(*a, **k)
        draws.append(out.clone())
        return out

    if profile == "long":
        # the scripted box (and the pads / tips placed relative to it) follows the position the reference's own reset drew,
        # so that the pre-grasp drift check (box vs initial_box_positions, 1 cm) can pass and stages 2 / 3 are reached
        pads_tips = model.fingerpad_local_indices + model.fingertip_local_indices

        def follow_box(env_, st):
            ib = env_.task.initial_box_positions.numpy()
            d = (ib[:, :2] - st["box"][:, :2]).astype(np.float32)
            st["box"][:, :2] += d
            for b in pads_tips:
                st["rbs"][:, b, :2] += d
            st["rbs"][:, -1, :] = st["box"]
        env.l1_hook = follow_box

    torch.rand = rec_rand
    rec = {k: [] for k in ("actions", "obs", "rew", "done", "targets", "active_prev_targets", "episode_step",
                           "reset_samples", "rew_total", "task_state", "stats")}
    l1_main, l1_reset, reset_l1_used = [], [], []
    ev = {k: [] for k in ("step", "ids", "samples", "l1", "targets", "active_prev_targets", "episode_step", "task_state",
                          "prev_actions_obs")}
    snaps = {"step": [], "oa": [], "rc": [], "rc_names": None}
    live_view_keys = set()

    def task_state_now():
        ts = env.observation_encoder.task_states
        return np.stack([ts["current_stage"].numpy().astype(np.float32), ts["time_in_stage"].numpy(),
                         ts["stage_contact_duration"].numpy(), ts["success_duration_steps"].numpy().astype(np.float32),
                         ts["just_transitioned_to_stage2"].numpy().astype(np.float32),
                         ts["just_transitioned_to_stage3"].numpy().astype(np.float32)])

    def collect_samples(env_ids_mask):
        """reshape the 6 rand calls of one reset_task_state into (N, 29) rows"""
        s = np.zeros((N, 29), dtype=np.float32)
        if has_box and draws:
            ids = np.nonzero(env_ids_mask)[0]
            x, y, yaw, tr, ro, fi = [d.numpy() for d in draws[-6:]]
            s[ids, 0], s[ids, 1], s[ids, 2] = x, y, yaw
            s[ids, 3:6], s[ids, 6:9], s[ids, 9:29] = tr, ro, fi
        draws.clear()
        return s

    try:
        # ---- env.reset(): reset all (consumes script state 0 as its physics step), obs x2
        c0 = env.l1_cursor
        obs0 = env.reset().clone()
        reset0_samples = collect_samples(np.ones(N, dtype=bool))
        reset0_l1 = pack_l1(script[c0], model, has_box)
        # a reset inside reset()'s post_physics_step would consume one more state
        extra_in_reset = env.l1_cursor - c0 - 1
        agen = np.random.default_rng(seed + 1)
        for t in range(T):
            if t in events:                                  # explicit reset_idx(ids) between two steps
                ids = np.asarray(events[t], dtype=np.int64)
                c = env.l1_cursor
                env.reset_idx(torch.as_tensor(ids))
                assert env.l1_cursor == c + 1                # ResetManager.reset_idx ran its physics step
                mask = np.zeros(N, dtype=bool)
                mask[ids] = True
                ev["step"].append(t)
                ev["ids"].append(np.pad(ids, (0, N - len(ids)), constant_values=-1))
                ev["samples"].append(collect_samples(mask))
                ev["l1"].append(pack_l1(script[c], model, has_box))
                ev["targets"].append(env.action_processor.full_dof_targets.numpy().copy())
                ev["active_prev_targets"].append(env.action_processor.active_prev_targets.numpy().copy())
                ev["episode_step"].append(env.episode_step_count.numpy().copy())
                ev["prev_actions_obs"].append(env.observation_encoder.prev_actions.numpy().copy())
                if has_box:
                    ev["task_state"].append(task_state_now())
            a = torch.tensor(2.0 * agen.random((N, env.num_actions), dtype=np.float32) - 1.0)
            c = env.l1_cursor
            env.obs_dict_pre_reset = None
            obs, rew, done, info = env.step(a)
            used = env.l1_cursor - c
            rec["actions"].append(a.numpy().copy())
            rec["obs"].append(obs.numpy().copy())
            rec["rew"].append(rew.numpy().copy())
            rec["done"].append(done.numpy().copy())
            rec["targets"].append(env.action_processor.full_dof_targets.numpy().copy())
            rec["active_prev_targets"].append(env.action_processor.active_prev_targets.numpy().copy())
            rec["episode_step"].append(env.episode_step_count.numpy().copy())
            rec["reset_samples"].append(collect_samples(done.numpy()))
            rec["rew_total"].append(info["reward_components"]["total"].numpy().copy())
            l1_main.append(pack_l1(script[c], model, has_box))
            l1_reset.append(pack_l1(script[c + 1], model, has_box) if used == 2 else None)
            reset_l1_used.append(used == 2)
            if has_box:
                ts = env.observation_encoder.task_states
                rec["task_state"].append(np.stack([ts["current_stage"].numpy().astype(np.float32),
                                                   ts["time_in_stage"].numpy(),
                                                   ts["stage_contact_duration"].numpy(),
                                                   ts["success_duration_steps"].numpy().astype(np.float32),
                                                   ts["just_transitioned_to_stage2"].numpy().astype(np.float32),
                                                   ts["just_transitioned_to_stage3"].numpy().astype(np.float32)]))
            if t in snap_steps:
                od_now = env.obs_dict_pre_reset if env.obs_dict_pre_reset is not None else env.obs_dict
                if env.obs_dict_pre_reset is not None:
                    live_view_keys.update(k for k in od_now if not torch.equal(od_now[k], env.obs_dict[k]))
                snaps["step"].append(t)
                snaps["oa"].append(np.concatenate([od_now[k].reshape(N, -1).numpy() for k, _ in OBS_KEYS if k in od_now], axis=1))
                snaps["rc"].append({k: v.numpy().copy() for k, v in info["reward_components"].items()})
            st = np.zeros(8, dtype=np.float32)
            st[0] = float(info["success_rate"])
            st[1] = float(info["failure_rate"])
            st[2] = float(info["timeout_rate"])
            st[3] = float(info["consecutive_successes"])
            rec["stats"].append(st)
    finally:
        torch.rand = real_rand

    out = {"N": N, "T": T, "task": task_name, "cfg_overrides": json.dumps(over),
           "obs0": obs0.numpy(), "reset0_samples": reset0_samples, "extra_in_reset": extra_in_reset,
           "reset_l1_used": np.array(reset_l1_used)}
    for k, v in rec.items():
        if v:
            out[k] = np.stack(v)
    if ev["step"]:
        out["ev_step"] = np.array(ev["step"])
        for k in ("ids", "samples", "targets", "active_prev_targets", "episode_step", "prev_actions_obs"):
            out[f"ev_{k}"] = np.stack(ev[k])
        if has_box:
            out["ev_task_state"] = np.stack(ev["task_state"])
        for f in ev["l1"][0]:
            out[f"ev_l1_{f}"] = np.stack([s_[f] for s_ in ev["l1"]])
    if snaps["step"]:
        out["snap_step"] = np.array(snaps["step"])
        out["snap_obs_all"] = np.stack(snaps["oa"])           # every obs_dict key, OBS_KEYS order
        out["snap_obs_keys"] = json.dumps([k for k, _ in OBS_KEYS if k in env.obs_dict])
        # keys whose obs_dict entry aliases an engine tensor in the reference (differs from the terminal observation after a
        # step with resets); recorded for INTEGRATION.md, not replayed
        out["snap_live_view_keys"] = json.dumps(sorted(live_view_keys))
        # the reference's dict does not carry every key at every step (e.g. before the first termination): union of the
        # names, NaN where a step's dict had no such key
        names = sorted(set().union(*[set(d) for d in snaps["rc"]]))
        out["snap_rc"] = np.stack([np.stack([d.get(k, np.full(N, np.nan, dtype=np.float32)) for k in names]) for d in snaps["rc"]])
        out["snap_rc_names"] = json.dumps(names)
    sparse_l1r = name in ("blind_wide", "base_wide", "blind_long")
    if sparse_l1r:
        idx, n_used = [], 0
        for s_ in l1_reset:
            idx.append(n_used if s_ is not None else -1)
            n_used += s_ is not None
        out["l1r_index"] = np.array(idx)
    fields = list(l1_main[0].keys())
    for f in fields:
        out[f"l1_{f}"] = np.stack([s[f] for s in l1_main])
        if sparse_l1r:   # new scenarios: the reset-step state only for the steps that had one (l1r_index: step -> row, -1 = none)
            used_rows = [s_[f] for s_ in l1_reset if s_ is not None]
            out[f"l1r_{f}"] = np.stack(used_rows) if used_rows else np.zeros((0,) + l1_main[0][f].shape, dtype=np.float32)
        else:
            out[f"l1r_{f}"] = np.stack([(s[f] if s is not None else np.zeros_like(l1_main[0][f])) for s in l1_reset])
        out[f"l1reset0_{f}"] = reset0_l1[f]
    # a few dictionary components to pin obs_all rows that are not in obs_buf
    od = env.obs_dict
    out["final_obs_dict_keys"] = json.dumps(sorted(od.keys()))
    for k in ("hand_pose_arr_aligned", "fingertip_poses_hand", "fingerpad_poses_hand", "contact_forces",
              "contact_force_magnitude", "all_finger_dof_vel", "all_finger_dof_target", "active_rule_targets"):
        out[f"final_{k}"] = od[k].reshape(N, -1).numpy().copy()
    if has_box:
        for k in ("finger_to_object_distances", "finger_to_object_height_diff", "hand_to_object_distance",
                  "grasp_state", "grasp_duration", "thumb_contact", "other_fingers_contact"):
            out[f"final_{k}"] = od[k].reshape(N, -1).numpy().copy()
    rc = env.last_reward_components
    out["final_reward_component_names"] = json.dumps(sorted(rc.keys()))
    for k, v in rc.items():
        out[f"final_rc_{k}"] = v.numpy().copy()
    return out


def pin_quaternion_standins():
    """Run the reference's own known-answer test against the stand-in torch_utils."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "ref_test_coordinate_transforms", os.path.join(REF, "dexhand_env/utils/test_coordinate_transforms.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import contextlib
    import io
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ok = mod.test_coordinate_transforms()
    assert ok is True, buf.getvalue()
    return buf.getvalue().count("PASSED")


def main():
    if not os.path.isdir(REF):
        print("reference not present; nothing generated")
        return 0
    install_standins()
    sys.path.insert(0, REF)
    n = pin_quaternion_standins()
    print(f"reference utils/test_coordinate_transforms.py: all cases passed against stand-ins ({n} PASSED lines)")
    # round-1 scenarios (N = 6..12) + round-2 scenarios: two workgroups incl. a padded one (N = 70), explicit reset_idx
    # events, per-step snapshots of every obs_dict key / reward component, and a default-length FSM run (stage 2, 3, success)
    table = [("blind_default", 8, 40, 11, None, ()), ("blind_fast", 12, 60, 12, None, ()),
             ("base_default", 6, 45, 13, None, ()), ("base_position", 6, 20, 14, None, ()),
             ("blind_wide", 70, 26, 15, {9: [3, 17, 65, 69], 20: [0, 64]}, (2, 6, 8, 13, 17, 21, 25)),
             ("base_wide", 70, 10, 16, {4: [1, 66]}, (0, 5, 9)),
             ("blind_long", 6, 345, 17, None, (197, 198, 199, 226, 227, 228, 327, 328, 344))]
    only = set(sys.argv[1:])
    for name, N, T, seed, events, snap in table:
        if only and name not in only:
            continue
        out = run_scenario(name, N, T, seed, events=events, snap_steps=snap)
        path = os.path.join(HERE, f"l2_{name}.npz")
        np.savez_compressed(path, **out)
        done = out["done"]
        print(f"{name}: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)  resets={int(done.sum())} "
              f"steps_with_reset={int(done.any(axis=1).sum())}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
