"""DexHandEnv: the reference's L3 env surface (DexHandBase / VecTask) over libdexsim.

Mirrors reference dexhand_env/tasks/dexhand_base.py (step :893-942, reset :805-838, reset_idx :743-803,
get_observations_dict :948-956, observation_space/action_space :1150-1169) and the attribute names its callers
read.  The heavy lifting is one C-ABI call per method; this file only holds views, dict
bookkeeping and the hooks for host-side custom rules.  There is no CPU path in the product: `sim_device='cpu'`
raises (the reference's CPU pipeline is itself documented as broken for multi-env, docs/guide-debugging.md:309-353).
"""
import copy

import numpy as np
import torch

from . import _abi
from .config import (FINGER_COUPLING_MAP, HARDWARE_MAPPING, OBS_KEYS, REWARD_TERMS, build_sim_config,
                     obs_key_offsets)
from .spaces import Box


class _PhysicsManagerView:
    """PhysicsManager attributes callers read (physics_manager.py:243-270)."""

    def __init__(self, sim_cfg):
        self.physics_dt = float(sim_cfg.dt)
        self.physics_steps_per_control_step = 2      # 1 main + 1 reset step, measured at init in the reference
        self.control_dt = float(sim_cfg.control_dt)
        self.auto_detected_physics_steps = True


class _ActionScalingView:
    """ActionScaling static helpers (scaling.py:28-99), torch-level, for callers that use them directly."""

    @staticmethod
    def scale_to_limits(actions, lower, upper):
        return (actions + 1.0) * 0.5 * (upper - lower) + lower

    @staticmethod
    def apply_velocity_deltas(prev_targets, actions, max_deltas):
        return prev_targets + actions * max_deltas

    @staticmethod
    def clamp_to_limits(targets, lower, upper):
        return torch.clamp(targets, lower, upper)

    @staticmethod
    def apply_velocity_clamp(new_targets, prev_targets, max_deltas):
        return prev_targets + torch.clamp(new_targets - prev_targets, -max_deltas, max_deltas)


class _ActionProcessorView:
    """ActionProcessor public API (action_processor.py:668-755) backed by the k_actions kernel."""

    def __init__(self, env):
        self._env = env
        c, dev = env._sim_cfg, env.device
        self.action_control_mode = "position" if c.control_mode == _abi.MODE_POSITION else "position_delta"
        self.policy_controls_hand_base = bool(c.policy_controls_base)
        self.policy_controls_fingers = bool(c.policy_controls_fingers)
        self.NUM_BASE_DOFS, self.NUM_ACTIVE_FINGER_DOFS = 6, 12
        self.finger_coupling_map = FINGER_COUPLING_MAP
        self.max_deltas = torch.tensor(list(c.max_deltas), device=dev)
        self.active_lower_limits = torch.tensor(list(c.active_lower), device=dev)
        self.active_upper_limits = torch.tensor(list(c.active_upper), device=dev)
        mask = torch.zeros(18, dtype=torch.bool, device=dev)
        mask[:6] = self.policy_controls_hand_base
        mask[6:] = self.policy_controls_fingers
        self.active_target_mask = mask
        self.action_scaling = _ActionScalingView()
        self._pre_action_rule = None
        self._action_rule = None
        # post-action filters (rules.py:113-135): registry + enabled list.  The two built-ins run inside the action
        # kernel; as soon as a custom filter is enabled or a custom coupling rule is set, the whole action stage runs
        # on the host in torch (this file) and only its results are handed to the device (slow path, see env.step).
        self._post_action_filter_registry = {
            "velocity_clamp": lambda prev, rule, tgt: prev + torch.clamp(tgt - prev, -self.max_deltas, self.max_deltas),
            "position_clamp": lambda prev, rule, tgt: torch.clamp(tgt, self.active_lower_limits, self.active_upper_limits),
        }
        self._enabled_post_action_filters = ["velocity_clamp", "position_clamp"]   # dexhand_base.py:478-481
        self._coupling_rule = None
        # coupling table as index / scale tensors for the host path (constants.py:71-88)
        name_to_dof = {n: i for i, n in enumerate(env.model.dof_names)}
        ctrl, dof, scale = [], [], []
        for f, lst in FINGER_COUPLING_MAP.items():
            for jn, sc in lst:
                ctrl.append(6 + f); dof.append(name_to_dof[jn]); scale.append(sc)
        self._cpl_ctrl = torch.tensor(ctrl, device=dev)
        self._cpl_dof = torch.tensor(dof, device=dev)
        self._cpl_scale = torch.tensor(scale, device=dev, dtype=torch.float32)

    def _host_path(self):
        """True when the action stage cannot run in the HIP kernel: custom filter enabled or custom coupling rule."""
        return self._coupling_rule is not None or self._enabled_post_action_filters != ["velocity_clamp", "position_clamp"]

    def _default_action_rule(self, prev, rule_targets, actions):
        """default_rules.py:33-112 (position / position_delta), torch restatement for the host path."""
        raw = rule_targets.clone()
        na = actions.shape[1]
        fs = 6 if self.policy_controls_hand_base else 0
        a = torch.zeros(actions.shape[0], 18, device=actions.device, dtype=actions.dtype)
        if self.policy_controls_hand_base:
            a[:, :6] = actions[:, :6]
        if self.policy_controls_fingers:
            a[:, 6:] = actions[:, fs:fs + 12] if na >= fs + 12 else 0.0
        m = self.active_target_mask
        if self.action_control_mode == "position_delta":
            raw[:, m] = prev[:, m] + a[:, m] * self.max_deltas[m]
            raw = torch.clamp(raw, self.active_lower_limits, self.active_upper_limits)
        else:
            lo, hi = self.active_lower_limits, self.active_upper_limits
            raw[:, m] = ((a + 1.0) * 0.5 * (hi - lo) + lo)[:, m]
        return raw

    def _host_process_actions(self, actions):
        """ActionProcessor.process_actions on the host (action_processor.py:284-352): rule -> enabled filters in order
        -> coupling; writes the results into the device state the kernels read."""
        core = self._env._core
        prev, rule_t = self.active_prev_targets.clone(), self.active_rule_targets.clone()
        if self._action_rule is not None:
            config = {"control_mode": self.action_control_mode, "policy_controls_base": self.policy_controls_hand_base,
                      "policy_controls_fingers": self.policy_controls_fingers}
            raw = self._action_rule(prev, rule_t, actions, config)
        else:
            raw = self._default_action_rule(prev, rule_t, actions)
        nxt = raw
        for name in self._enabled_post_action_filters:
            fn = self._post_action_filter_registry.get(name)
            if fn is not None:                                   # rules.py:122-133: unknown names are skipped
                nxt = fn(prev, rule_t, nxt)
        if self._coupling_rule is not None:
            full = self._coupling_rule(nxt)
        else:
            full = torch.zeros(nxt.shape[0], 26, device=nxt.device, dtype=nxt.dtype)
            full[:, :6] = nxt[:, :6]
            full[:, self._cpl_dof] = nxt[:, self._cpl_ctrl] * self._cpl_scale
        core.field("active_prev_targets").copy_(nxt.t())
        core.field("targets").copy_(full.t())
        core.full_dof_targets.copy_(full)
        a18 = torch.zeros(actions.shape[0], 18, device=actions.device, dtype=actions.dtype)
        a18[:, :actions.shape[1]] = actions
        core.field("actions").copy_(a18.t())
        core.field("prev_actions").copy_(a18.t())
        core.begin_step()

    @property
    def active_prev_targets(self):
        return self._env._core.field("active_prev_targets").t()

    @property
    def active_rule_targets(self):
        return self._env._core.field("active_rule_targets").t()

    @property
    def full_dof_targets(self):
        return self._env._core.full_dof_targets

    @property
    def control_dt(self):
        return self._env.physics_manager.control_dt

    def set_pre_action_rule(self, rule):
        """rule(active_prev_targets (N,18), state={'obs_dict','env'}) -> active_rule_targets (rules.py:78-95)."""
        self._pre_action_rule = rule

    def set_action_rule(self, rule):
        """rule(active_prev_targets, active_rule_targets, actions, config) -> active_raw_targets (rules.py:97-111).
        Runs on the host in torch; the velocity/position clamps and the coupling still run in the HIP kernel."""
        self._action_rule = rule
        if rule is None:
            self._env._core.set_raw_targets(None)

    def register_post_action_filter(self, name, filter_fn):
        """filter_fn(active_prev_targets, active_rule_targets, active_targets) -> filtered (rules.py:113-135).  Like in the
        reference, registering does not enable: the name must also be in `_enabled_post_action_filters`
        (dexhand_base.py:255-261 extends that list with the task's `post_action_filters`)."""
        self._post_action_filter_registry[name] = filter_fn

    def set_coupling_rule(self, rule):
        """rule(active_targets (N,18)) -> full_dof_targets (N,26) (action_processor.py:698-705); None restores the table."""
        self._coupling_rule = rule

    def unscale_actions(self, actions):
        """actions in [-1, 1] -> physical units (action_processor.py:721-755)."""
        if self.action_control_mode == "position":
            return _ActionScalingView.scale_to_limits(actions, self.active_lower_limits[self.active_target_mask],
                                                      self.active_upper_limits[self.active_target_mask])
        return actions * self.max_deltas[self.active_target_mask]


class _ObservationEncoderView:
    """The obs_dict accessors callers use (observation_encoder.py:999-1665, the live subset)."""

    def __init__(self, env):
        self._env = env
        self.observation_keys = list(env.task_cfg["policy_observation_keys"])
        self.num_observations = int(env._sim_cfg.num_obs)
        self.raw_dof_name_to_index = {n: i for i, n in enumerate(env.model.dof_names)}
        self.control_name_to_index = {n: i for i, (n, _) in enumerate(HARDWARE_MAPPING)}
        idx, slices = 0, {}
        for k, (off, ln) in zip(self.observation_keys, env._obs_segments):
            slices[k] = (idx, idx + ln)
            idx += ln
        self.component_slice_indices = slices

    @property
    def task_states(self):
        c = self._env._core
        names = ["success_duration_steps", "success_conditions_met", "current_stage", "time_in_stage",
                 "stage_contact_duration", "just2", "just3"]
        out = {n: c.field(n)[0] for n in names}
        out["just_transitioned_to_stage2"] = out.pop("just2")
        out["just_transitioned_to_stage3"] = out.pop("just3")
        return out

    @property
    def obs_buf(self):
        return self._env.obs_buf

    def get_raw_finger_dof(self, dof_name, obs_type="pos", obs_data=None, env_idx=None):
        if dof_name not in self.raw_dof_name_to_index:
            raise ValueError(f"Unknown DOF name: {dof_name}. Available: {list(self.raw_dof_name_to_index.keys())}")
        if obs_data is None:
            raise ValueError("obs_data must be provided")
        key = {"pos": "all_finger_dof_pos", "vel": "all_finger_dof_vel", "target": "all_finger_dof_target"}.get(obs_type)
        if key is None:
            raise ValueError(f"Unknown obs_type: {obs_type}. Available: pos, vel, target")
        if not isinstance(obs_data, dict):
            raise ValueError(f"Raw finger DOF '{dof_name}' access requires obs_dict.")
        i = self.raw_dof_name_to_index[dof_name] - 6
        if i < 0:
            raise ValueError(f"DOF {dof_name} is not a finger DOF")
        data = obs_data[key][:, i]
        return data[env_idx].item() if env_idx is not None else data


class _PolicyObsDict(dict):
    """obs_dict in mode "policy" (env.obsDict: policy): the policy keys are views of obs_buf; every other key of the reference's
    obs_dict exists in the reference but is not materialised here -- asking for one is an error that says how to get it."""

    def __init__(self, views, all_keys):
        super().__init__(views)
        self._all_keys = set(all_keys)

    def copy(self):
        return _PolicyObsDict(dict(self), self._all_keys)

    def __missing__(self, key):
        if key in self._all_keys:
            raise KeyError(f"obs_dict['{key}'] is not materialised in env.obsDict = 'policy' mode (only the policy_observation_keys "
                           f"are: {sorted(self)}); create the env with cfg['env']['obsDict'] = 'all' (the default) to read it")
        raise KeyError(key)


class _TerminationManagerView:
    def __init__(self, env):
        self._env = env
        self.max_episode_length = int(env._sim_cfg.episode_length)
        self.max_consecutive_successes = int(env._sim_cfg.max_consecutive_successes)

    @property
    def consecutive_successes(self):
        return self._env._core.stats[_abi.STAT["CONSECUTIVE_SUCCESSES"]]

    @property
    def episode_success(self):
        return self._env._core.field("episode_success")[0].bool()

    @property
    def episode_failure(self):
        return self._env._core.field("episode_failure")[0].bool()


class _TaskView:
    def __init__(self, name):
        self.name = name


class DexHandEnv:
    """Vectorised DexHand environment on one MI355X (one process per GPU; shard envs across ranks)."""

    def __init__(self, cfg, task_name, rl_device, sim_device, graphics_device_id=0, headless=True, force_render=False,
                 video_config=None, domain_randomisation=None, hand_model=None, _core_factory=None):
        self.cfg = cfg
        self.env_cfg, self.task_cfg, self.sim_cfg = cfg["env"], cfg["task"], cfg["sim"]
        self.video_config = video_config
        self.headless, self.force_render = headless, force_render
        self.graphics_device_id = graphics_device_id
        # Env.__init__ device rule (vec_task.py:61-84): "cuda:k"/"gpu" => GPU pipeline, anything else => cpu
        split = str(sim_device).split(":")
        dev_type = split[0].lower()
        if "use_gpu_pipeline" in cfg.get("sim", {}):
            raise RuntimeError("The 'use_gpu_pipeline' config key is deprecated and must be removed. "
                               "GPU pipeline is now automatically determined from sim_device.")
        if dev_type in ("cuda", "gpu"):
            self.use_gpu_pipeline = True
            self.device = f"cuda:{int(split[1]) if len(split) > 1 else 0}"
        else:
            self.use_gpu_pipeline = False
            self.device = "cpu"
        self.rl_device = rl_device
        if torch.device(rl_device) != torch.device(self.device) and _core_factory is None:
            # TensorManager refuses wrapped tensors that are not on rl_device (tensor_manager.py:178-185)
            raise RuntimeError(f"Device mismatch: rl_device '{rl_device}' but simulation tensors live on '{self.device}'.")
        self.max_episode_length = self.env_cfg["episodeLength"]
        self.num_envs = int(self.env_cfg["numEnvs"])
        self.clip_obs = self.env_cfg.get("clipObservations", np.inf)      # read, never applied (vec_task.py:108-109)
        self.clip_actions = self.env_cfg.get("clipActions", np.inf)
        self.seed(cfg["train"]["seed"])

        # hand_model: a HandModel, e.g. dexrobot_isaac_amd.mjcf.load_mjcf(<dexhand021_right_simplified_floating.xml>);
        # None = the authored stand-in (the reference's MJCF submodule is absent offline)
        self._sim_cfg, self.model = build_sim_config(cfg, model=hand_model, dr=domain_randomisation)
        self._model_struct = self.model.to_struct()
        if _core_factory is None:
            from .core import DexSimCore
            if not self.use_gpu_pipeline:
                from ._lib import DexSimError
                raise DexSimError("sim_device='cpu' is not supported: dexsim is a HIP-only engine with no CPU fallback")
            self._core = DexSimCore(self._sim_cfg, self._model_struct, self.device)
        else:
            self._core = _core_factory(self._sim_cfg, self._model_struct, self.device)
        core = self._core
        self.task = _TaskView(task_name)
        self.physics_dt = float(self._sim_cfg.dt)
        self.dt = self.physics_dt
        self.physics_manager = _PhysicsManagerView(self._sim_cfg)
        self.physics_steps_per_control_step = 2
        self._num_observations = int(self._sim_cfg.num_obs)
        self._num_actions = int(self._sim_cfg.num_actions)
        self.num_states = 0
        self.num_dof = _abi.NJ
        self._obs_segments = [(int(self._sim_cfg.obs_seg_off[i]), int(self._sim_cfg.obs_seg_len[i]))
                              for i in range(int(self._sim_cfg.n_obs_seg))]
        # index contract the reference resolves through Isaac Gym name lookups (hand_initializer.py:439-588)
        self.base_joint_names = self.model.dof_names[:6]
        self.finger_joint_names = [n for n in self.model.dof_names[6:] if n != "r_f_joint3_1"]
        self.hand_local_rigid_body_index = self.model.hand_local_rigid_body_index
        self.hand_local_actor_index = 0
        self.fingertip_local_indices = self.model.fingertip_local_indices
        self.fingerpad_local_indices = self.model.fingerpad_local_indices
        self.contact_force_local_body_indices = self.model.body_indices(self.task_cfg["contact_force_bodies"])
        self.dof_props = torch.tensor(self.model.dof_props(), device=self.device)
        self.action_control_mode = self.task_cfg["controlMode"]
        self.policy_controls_hand_base = bool(self.task_cfg["policy_controls_hand_base"])
        self.policy_controls_fingers = bool(self.task_cfg["policy_controls_fingers"])
        self.action_processor = _ActionProcessorView(self)
        self.observation_encoder = _ObservationEncoderView(self)
        self.termination_manager = _TerminationManagerView(self)

        # buffers (initialization_manager.py:34-117): env-owned, returned by reference from step()
        self.obs_buf, self.states_buf = core.obs_buf, core.obs_buf
        self.rew_buf, self.reset_buf = core.rew_buf, core.reset_buf
        self.episode_step_count = core.episode_step_count
        self.actions = torch.zeros((self.num_envs, self._num_actions), device=self.device)
        self._actions_copy_bound = hasattr(core, "set_action_copy")     # (the CPU oracle stand-in of the tests has no sink)
        if self._actions_copy_bound:
            core.set_action_copy(self.actions)
        self.dof_state = core.dof_state
        self.dof_pos, self.dof_vel = core.dof_state[..., 0], core.dof_state[..., 1]
        self.actor_root_state_tensor = core.root_state
        self._build_views()

        # control-cycle measurement of the reference's __init__ (dexhand_base.py:270-320): zero targets,
        # one physics step, reset of every env (with its physics step); control_dt = 2 * sim.dt
        core.process_actions(self.actions, zero_targets=True)
        core.physics_step()
        core.reset_idx(torch.arange(self.num_envs, device=self.device))
        self._apply_pre_action_rule()
        self._initialization_complete = True

    # ------------------------------------------------------------------ construction helpers
    def seed(self, seed=None):                                   # vec_task.py:145-153
        if seed is None:
            return
        import random
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)

    def _build_views(self):
        core, c = self._core, self._sim_cfg
        offs = obs_key_offsets()
        oa = core.field("obs_all")
        n_keys = len(OBS_KEYS) if c.task == _abi.TASK_BLIND_GRASPING else 23
        # env.obsDict (new key): "all" (default) = every key of the reference's obs_dict is materialised each step (392 SoA rows per
        # env); "policy" = only obs_buf is written and obs_dict serves the policy keys as views of it -- what a training loop
        # needs (dexsim_set_obs_dict_mode; 1 568 B per env and step less to write)
        self.obs_dict_mode = str(self.env_cfg.get("obsDict", "all"))
        if self.obs_dict_mode not in ("all", "policy"):
            raise ValueError(f"env.obsDict must be 'all' or 'policy', got '{self.obs_dict_mode}'")
        if self.obs_dict_mode == "policy" and hasattr(core, "set_obs_dict_mode"):
            core.set_obs_dict_mode(True)
            views = {k: self.obs_buf[:, a:b] for k, (a, b) in self.observation_encoder.component_slice_indices.items()}
            self.obs_dict = _PolicyObsDict(views, [k for k, _ in OBS_KEYS[:n_keys]])
        else:
            self.obs_dict_mode = "all"
            self.obs_dict = {}
        for name, dim in OBS_KEYS[:n_keys]:
            if self.obs_dict_mode != "all":
                break
            off, _ = offs[name]
            if name == "prev_actions":
                dim = int(c.num_actions)
            self.obs_dict[name] = oa[off:off + dim].t()          # (N, dim) view of the SoA rows
        rc = core.field("rew_comp")
        comps = {}
        for i, name in enumerate(REWARD_TERMS):
            if float(c.reward_weight[i]) != 0.0:                  # reward_calculator.py:255-270
                comps[name] = rc[i]
                comps[f"{name}_weighted"] = rc[_abi.REWROW_WEIGHTED + i]
        comps["total"] = rc[_abi.REWROW_TOTAL]
        for j, name in enumerate(("success", "failure_penalty", "timeout_penalty")):   # step_processor.py:204-219
            comps[f"termination_{name}"] = rc[_abi.REWROW_TERM_RAW + j]
            comps[f"termination_{name}_weighted"] = rc[_abi.REWROW_TERM_W + j]
        self.last_reward_components = comps
        st, mk = core.stats, core.masks
        ex = {"consecutive_successes": st[_abi.STAT["CONSECUTIVE_SUCCESSES"]], "episode_length": core.episode_length}
        for i, name in enumerate(_abi.SUCCESS_CRITERIA):
            if (c.active_success_mask >> i) & 1:
                ex[f"success_{name}"] = st[_abi.STAT["SUCC_MEAN"] + i]
                ex[f"success_reason_{name}"] = mk[_abi.MASK_SUCC_REASON + i]
        for i, name in enumerate(_abi.FAILURE_CRITERIA):
            if (c.active_failure_mask >> i) & 1:
                ex[f"failure_{name}"] = st[_abi.STAT["FAIL_MEAN"] + i]
                ex[f"failure_reason_{name}"] = mk[_abi.MASK_FAIL_REASON + i]
        ex["success"], ex["failure"], ex["timeout"] = mk[_abi.MASK_SUCCESS], mk[_abi.MASK_FAILURE], mk[_abi.MASK_TIMEOUT]
        ex["success_rate"] = st[_abi.STAT["SUCCESS_RATE"]]
        ex["failure_rate"] = st[_abi.STAT["FAILURE_RATE"]]
        ex["timeout_rate"] = st[_abi.STAT["TIMEOUT_RATE"]]
        ex["reward_components"] = comps
        self.extras = ex

    # ------------------------------------------------------------------ VecTask properties
    @property
    def num_observations(self):
        if self._num_observations == 0:
            raise RuntimeError("num_observations accessed before initialization.")
        return self._num_observations

    @property
    def num_actions(self):
        if self._num_actions == 0:
            raise RuntimeError("num_actions accessed before initialization.")
        return self._num_actions

    @property
    def observation_space(self):
        return Box(-float("inf"), float("inf"), (self.num_observations,), np.float32)

    @property
    def action_space(self):
        return Box(-1.0, 1.0, (self.num_actions,), np.float32)

    @property
    def episode_time(self):                                      # dexhand_base.py:702-712
        return self.episode_step_count.float() * self.physics_manager.control_dt

    @property
    def random_actions_enabled(self):                            # viewer-only feature, headless => False
        return False

    @property
    def rigid_body_states(self):
        """(N, B, 13), materialised on access (gym.refresh_rigid_body_state_tensor)."""
        self._core.refresh_body_states()
        return self._core.rigid_body_states

    @property
    def contact_forces_all(self):
        self._core.refresh_body_states()
        return self._core.contact_forces_all

    @property
    def contact_forces(self):
        """(N, 5, 3) forces on r_f_link{1..5}_4, the per-step gathered copy (tensor_manager.py:441-446)."""
        return self._core.field("cf5").t().reshape(self.num_envs, 5, 3)

    @property
    def full_dof_targets(self):
        return self._core.full_dof_targets

    # ------------------------------------------------------------------ lifecycle
    def _apply_pre_action_rule(self):
        """The custom pre-action rule (rules.py:78-95).  The reference applies it between compute_observations and
        concatenate_observations (step_processor.py:56-77), i.e. on the obs_dict of the terminal state, before termination,
        rewards and the in-step resets.  Here the fused step has already run all of that with the identity default; nothing
        in termination or rewards reads active_rule_targets, and obs_all still holds the PRE-reset observations, so the
        rule is evaluated afterwards on the same inputs the reference gives it -- obs_dict of the terminal state and the
        pre-reset active_prev_targets (the obs_dict entry, not the post-reset field) -- and its output is patched into
        the field the next action stage reads, the obs_dict row and, when the key is a policy observation, into the
        obs_buf / rollout-sink columns."""
        rule = self.action_processor._pre_action_rule
        if rule is None:       # identity default already written by the post kernel
            return
        if self.obs_dict_mode != "all":
            raise RuntimeError("a custom pre-action rule reads the full obs_dict: create the env with env.obsDict = 'all' (the default)")
        out = rule(self.obs_dict["active_prev_targets"].clone(), {"obs_dict": self.obs_dict, "env": self})
        self._core.field("active_rule_targets").copy_(out.t())
        off, dim = obs_key_offsets()["active_rule_targets"]
        self._core.field("obs_all")[off:off + dim].copy_(out.t())
        sl = self.observation_encoder.component_slice_indices.get("active_rule_targets")
        if sl is not None:
            self.obs_buf[:, sl[0]:sl[1]] = out
            sink = getattr(self._core, "_sink_obs", None)
            if sink is not None:
                sink[:, sl[0]:sl[1]] = out

    def step(self, actions):
        """obs (N,O) f32, rew (N,) f32, done (N,) bool, extras -- views of env-owned buffers."""
        if actions is None:
            raise RuntimeError("Actions cannot be None")
        ap = self.action_processor
        fused = not ap._host_path() and ap._action_rule is None
        if fused and self._actions_copy_bound:
            # DexHandBase.actions = actions.clone() (dexhand_base.py:851): the copy is written by the action block of the
            # step kernel itself (dexsim_set_action_copy) into self.actions -- no clone launch on the hot path
            self._core.step(actions)
            self._apply_pre_action_rule()
            return self.obs_buf, self.rew_buf, self.reset_buf, self.extras
        if self._actions_copy_bound:
            self.actions.copy_(actions)      # keep the tensor the device copy is bound to
        else:
            self.actions = actions.clone()                       # dexhand_base.py:851
        if ap._host_path():
            # custom post-action filter / coupling rule: action stage in torch on the host, then the staged device path
            ap._host_process_actions(self.actions)
            self._core.physics_step(False)
            self._core.post_physics(False)
            self._apply_pre_action_rule()
            return self.obs_buf, self.rew_buf, self.reset_buf, self.extras
        if ap._action_rule is not None:
            config = {"control_mode": ap.action_control_mode, "policy_controls_base": ap.policy_controls_hand_base,
                      "policy_controls_fingers": ap.policy_controls_fingers}
            raw = ap._action_rule(ap.active_prev_targets.clone(), ap.active_rule_targets.clone(), self.actions, config)
            self._core.set_raw_targets(raw)
        self._core.step(self.actions)
        self._apply_pre_action_rule()
        return self.obs_buf, self.rew_buf, self.reset_buf, self.extras

    def reset(self):
        if self.action_processor._pre_action_rule is None:
            self._core.reset()
            return self.obs_buf
        # with a custom pre-action rule the reference's sequence is spelled out, because the rule runs inside BOTH
        # observation passes of reset() (dexhand_base.py:805-838: compute_observations, then post_physics_step)
        core = self._core
        core.begin_step()
        core.reset_idx(torch.arange(self.num_envs, device=self.device))
        core.post_physics(True)
        self._apply_pre_action_rule()
        core.post_physics(False)
        self._apply_pre_action_rule()
        return self.obs_buf

    def reset_idx(self, env_ids):
        if len(env_ids) == 0:
            return
        self._core.reset_idx(env_ids)

    def pre_physics_step(self, actions):
        if self._actions_copy_bound:
            self.actions.copy_(actions)      # keep the tensor the device-side action copy is bound to
        else:
            self.actions = actions.clone()   # dexhand_base.py:851
        self._core.process_actions(self.actions)

    def post_physics_step(self):
        self._core.post_physics(False)
        return self.obs_buf, self.rew_buf, self.reset_buf, self.extras

    def get_observations_dict(self):
        return self.obs_dict.copy()

    def set_rule_based_controllers(self, base_controller=None, finger_controller=None):
        """Rule-based control of the DOF groups the policy does not drive (dexhand_base.py:958-986), realised as a
        pre-action rule that overwrites the uncontrolled part of active_rule_targets."""
        def rule(active_prev_targets, state):
            out = active_prev_targets.clone()
            if base_controller is not None and not self.policy_controls_hand_base:
                out[:, :6] = base_controller(self)
            if finger_controller is not None and not self.policy_controls_fingers:
                out[:, 6:] = finger_controller(self)
            return out
        self.action_processor.set_pre_action_rule(rule if (base_controller or finger_controller) else None)

    def render(self, mode="rgb_array"):
        return None                                              # headless: no viewer / recorder / streamer

    def close(self):
        if self._core is not None:
            self._core.close()

    # rl_games-side expectations (rl/__init__.py:39-59)
    def get_env_info(self):
        return {"action_space": self.action_space, "observation_space": self.observation_space, "num_envs": self.num_envs}

    def get_number_of_agents(self):
        return 1
