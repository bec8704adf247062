"""OracleCore: a DexSimCore look-alike backed by the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Lets the `-m "not gpu"` tests exercise the product's host logic (DexHandEnv, factory, rollout buffer, the
world_size-2 gloo path) without a GPU by injecting it through DexHandEnv(_core_factory=...).  The product never
imports this module."""
import numpy as np
import torch

from dexrobot_isaac_amd import _abi
from oracle.oracle import Oracle

_INT_FIELDS = {"ncontact", "episode_step", "success_duration_steps", "success_conditions_met", "current_stage", "just2",
               "just3", "prev_contacts", "episode_success", "episode_failure", "success_reason", "failure_reason",
               "crit_success", "crit_failure", "term_success", "term_failure", "term_timeout", "reset_flag", "reset_count"}


class OracleCore:
    def __init__(self, sim_cfg, model_struct, device="cpu", threads=0):
        self.cfg, self.model = sim_cfg, model_struct
        self.device = torch.device("cpu")
        self.N = int(sim_cfg.num_envs)
        self.orc = Oracle(sim_cfg, model_struct, threads=threads)
        N = self.N
        A, B = 1 + int(sim_cfg.has_box), _abi.NUM_HAND_BODIES + int(sim_cfg.has_box)
        self.num_bodies, self.num_actors = B, A
        self.stats = torch.zeros(_abi.STAT_WORDS)
        self.obs_buf = torch.zeros(N, sim_cfg.num_obs)
        self.rew_buf = torch.zeros(N)
        self.reset_buf = torch.zeros(N, dtype=torch.bool)
        self.episode_step_count = torch.zeros(N, dtype=torch.int64)
        self.episode_length = torch.zeros(N, dtype=torch.int64)
        self.dof_state = torch.zeros(N, _abi.NJ, 2)
        self.root_state = torch.zeros(N, A, 13)
        self.rigid_body_states = torch.zeros(N, B, 13)
        self.contact_forces_all = torch.zeros(N, B, 3)
        self.full_dof_targets = torch.zeros(N, _abi.NJ)
        self.masks = torch.zeros(_abi.NUM_MASKS, N, dtype=torch.bool)
        self._fields = {}
        self._raw = None
        self._sink = (None, None, None)
        self._sink_stats = None
        self._sync()

    # persistent tensors so that views handed out by DexHandEnv stay valid (updated in place)
    def field(self, name):
        if name not in self._fields:
            v = self.orc.get(name)
            self._fields[name] = torch.as_tensor(v, dtype=torch.int32 if name in _INT_FIELDS else torch.float32).clone()
        return self._fields[name]

    def _sync(self):
        o = self.orc
        for name, t in self._fields.items():
            t.copy_(torch.as_tensor(o.get(name)).to(t.dtype))
        self.obs_buf.copy_(torch.as_tensor(o.obs_buf()))
        self.rew_buf.copy_(torch.as_tensor(o.get("rew")[0], dtype=torch.float32))
        self.reset_buf.copy_(torch.as_tensor(o.get("reset_flag")[0] != 0))
        es = torch.as_tensor(o.get("episode_step")[0]).to(torch.int64)
        self.episode_step_count.copy_(es)
        self.stats.copy_(torch.as_tensor(o.stats()))
        self.dof_state[..., 0] = torch.as_tensor(o.get("q").T, dtype=torch.float32)
        self.dof_state[..., 1] = torch.as_tensor(o.get("qd").T, dtype=torch.float32)
        self.full_dof_targets.copy_(torch.as_tensor(o.get("targets").T, dtype=torch.float32))
        if self.cfg.has_box:
            self.root_state[:, 1, 0:3] = torch.as_tensor(o.get("box_pos").T, dtype=torch.float32)
            self.root_state[:, 1, 3:7] = torch.as_tensor(o.get("box_quat").T, dtype=torch.float32)
            self.root_state[:, 1, 7:10] = torch.as_tensor(o.get("box_lin").T, dtype=torch.float32)
            self.root_state[:, 1, 10:13] = torch.as_tensor(o.get("box_ang").T, dtype=torch.float32)
        m = self.masks
        m[_abi.MASK_SUCCESS] = torch.as_tensor(o.get("term_success")[0] != 0)
        m[_abi.MASK_FAILURE] = torch.as_tensor(o.get("term_failure")[0] != 0)
        m[_abi.MASK_TIMEOUT] = torch.as_tensor(o.get("term_timeout")[0] != 0)
        m[_abi.MASK_SUCC_REASON:_abi.MASK_SUCC_REASON + 1] = torch.as_tensor(o.get("success_reason") != 0)
        m[_abi.MASK_FAIL_REASON:_abi.MASK_FAIL_REASON + 5] = torch.as_tensor(o.get("failure_reason") != 0)

    def set_reset_samples(self, s):
        self.orc.set_reset_samples(None if s is None else np.asarray(s, dtype=np.float32))

    def set_raw_targets(self, raw):
        if raw is not None:
            raise NotImplementedError("custom action rules are exercised on the GPU path only")

    def init_state(self):
        self.orc.init_state()
        self._sync()

    def begin_step(self):
        pass                                   # device-side step bookkeeping only exists on the HIP path

    def process_actions(self, actions, zero_targets=False):
        self.orc.process_actions(actions.detach().cpu().numpy(), zero_targets)
        self._sync()

    def physics_step(self, gate_on_reset=False):
        self.orc.physics_step()
        self._sync()

    def post_physics(self, obs_only=False):
        self.orc.post_physics(obs_only)
        self._finish_step()

    def _finish_step(self, sink=False):
        self._sync()
        self.episode_length.copy_(self.episode_step_count)
        if sink:                               # what the HIP step's flush / closing launch do (dexsim_set_step_sink, dexsim_set_stats_sink)
            so, sr, sd = self._sink
            if so is not None:
                so.copy_(self.obs_buf)
            if sr is not None:
                sr.copy_(self.rew_buf)
            if sd is not None:
                sd.copy_(self.reset_buf.to(torch.uint8))
            if self._sink_stats is not None:
                self._sink_stats[: _abi.STAT_USED].copy_(self.stats[: _abi.STAT_USED])

    def set_step_sink(self, obs=None, rew=None, done=None):
        self._sink = (obs, rew, done)
        self._sink_obs = obs

    def set_stats_sink(self, dst):
        self._sink_stats = dst

    def step(self, actions):
        a = actions.detach().cpu().numpy().astype(np.float32)
        self.orc.lib.orc_step(self.orc.h, np.ascontiguousarray(a).ctypes.data)
        # obs/rew/done of the oracle are the pre-reset values of the step, like the reference's returned buffers
        self._finish_step(sink=True)
        # episode_length is cloned before... after the resets in the reference (step_processor.py:221-232)

    def reset(self):
        self.orc.reset()
        self._finish_step()

    def reset_idx(self, env_ids):
        ids = torch.as_tensor(env_ids).detach().cpu().numpy().astype(np.int64)
        if len(ids):
            self.orc.reset_idx(ids)
            self._sync()

    def refresh_body_states(self):
        raise NotImplementedError("full rigid_body_states materialisation is a HIP-path feature")

    def close(self):
        pass
