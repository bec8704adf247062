"""DexSimCore: device memory + streams plumbing around the C-ABI (PyTorch-ROCm is used only for allocation,
the current stream and tensor views; every computation is a HIP kernel launched by libdexsim)."""
import ctypes as C

import torch

from . import _abi
from ._lib import DexSimError, check, load


class DexSimCore:
    def __init__(self, sim_cfg, model_struct, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise DexSimError(f"dexsim needs a HIP device (got '{device}'): there is no CPU backend")
        if not torch.cuda.is_available():
            raise DexSimError("no HIP device visible to PyTorch-ROCm")
        self.lib = load()
        self.cfg = sim_cfg
        self.model = model_struct
        self.device = device
        self.N = int(sim_cfg.num_envs)
        self.NS = (self.N + 63) // 64 * 64
        dev_index = device.index if device.index is not None else torch.cuda.current_device()
        self.dev_index = dev_index

        fields = (_abi.DexSimField * 128)()
        nf, words = C.c_int(0), C.c_size_t(0)
        check(self.lib.dexsim_arena_layout(C.byref(sim_cfg), fields, 128, C.byref(nf), C.byref(words)), "arena_layout")
        self.arena = torch.zeros(words.value, dtype=torch.float32, device=device)
        self._arena_i32 = self.arena.view(torch.int32)
        self.fields = {}
        for i in range(nf.value):
            f = fields[i]
            self.fields[f.name.decode()] = (int(f.offset), int(f.rows), bool(f.is_int))

        N = self.N
        A = 1 + int(sim_cfg.has_box)
        B = _abi.NUM_HAND_BODIES + int(sim_cfg.has_box)
        self.num_bodies, self.num_actors = B, A
        z = lambda *s, dtype=torch.float32: torch.zeros(*s, dtype=dtype, device=device)
        self.stats = z(_abi.STAT_WORDS)
        self.counters = z(_abi.STAT_WORDS, dtype=torch.int32)
        self.obs_buf = z(N, sim_cfg.num_obs)
        self.rew_buf = z(N)
        self.reset_buf = z(N, dtype=torch.bool)
        self.episode_step_count = z(N, dtype=torch.int64)
        self.episode_length = z(N, dtype=torch.int64)
        self.dof_state = z(N, _abi.NJ, 2)
        self.root_state = z(N, A, 13)
        self.rigid_body_states = z(N, B, 13)
        self.contact_forces_all = z(N, B, 3)
        self.full_dof_targets = z(N, _abi.NJ)
        self.reset_samples = None
        self.masks = z(_abi.NUM_MASKS, N, dtype=torch.bool)
        self.raw_targets = None

        self.h = C.c_void_p()
        check(self.lib.dexsim_create(C.byref(sim_cfg), C.byref(model_struct), dev_index, C.byref(self.h)), "create")
        self._bind()
        self.init_state()

    # ------------------------------------------------------------------ plumbing
    def _bind(self):
        b = _abi.DexSimBuffers()
        b.arena = self.arena.data_ptr()
        b.stats, b.counters = self.stats.data_ptr(), self.counters.data_ptr()
        b.obs_buf, b.rew_buf, b.reset_buf = self.obs_buf.data_ptr(), self.rew_buf.data_ptr(), self.reset_buf.data_ptr()
        b.episode_step_count, b.episode_length = self.episode_step_count.data_ptr(), self.episode_length.data_ptr()
        b.dof_state, b.root_state = self.dof_state.data_ptr(), self.root_state.data_ptr()
        b.rigid_body_states, b.contact_forces_all = self.rigid_body_states.data_ptr(), self.contact_forces_all.data_ptr()
        b.full_dof_targets = self.full_dof_targets.data_ptr()
        b.reset_samples = self.reset_samples.data_ptr() if self.reset_samples is not None else None
        b.masks = self.masks.data_ptr()
        b.raw_targets = self.raw_targets.data_ptr() if self.raw_targets is not None else None
        check(self.lib.dexsim_bind(self.h, C.byref(b)), "bind")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.dexsim_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def field(self, name):
        """(rows, N) view of an arena field (float32 or int32); padded lanes are sliced off."""
        off, rows, is_int = self.fields[name]
        base = self._arena_i32 if is_int else self.arena
        return base[off: off + rows * self.NS].view(rows, self.NS)[:, : self.N]

    def set_reset_samples(self, samples):
        """Inject the 29 uniforms/env of the next resets (parity tests); None -> device Philox stream."""
        if samples is None:
            self.reset_samples = None
        else:
            s = torch.as_tensor(samples, dtype=torch.float32, device=self.device).contiguous()
            assert s.shape == (self.N, _abi.NRESET_SAMPLES)
            self.reset_samples = s
        self._bind()

    def set_raw_targets(self, raw):
        """(N, 18) output of a host-side custom action rule for the next process_actions; None -> built-in rule."""
        if raw is None:
            if self.raw_targets is not None:
                self.raw_targets = None
                self._bind()
            return
        first = self.raw_targets is None
        if first:
            self.raw_targets = torch.zeros(self.N, _abi.NACT, device=self.device)
        self.raw_targets.copy_(raw)
        if first:
            self._bind()

    def _actions_ptr(self, actions):
        if actions is None:
            raise RuntimeError("Actions cannot be None")
        if actions.device != self.device or actions.dtype != torch.float32 or not actions.is_contiguous():
            actions = actions.to(device=self.device, dtype=torch.float32).contiguous()
        if tuple(actions.shape) != (self.N, int(self.cfg.num_actions)):
            raise DexSimError(f"actions must have shape ({self.N}, {int(self.cfg.num_actions)}), got {tuple(actions.shape)}")
        self._keep = actions  # keep alive until the kernel ran
        return C.c_void_p(actions.data_ptr())

    def _ids_ptr(self, env_ids):
        ids = torch.as_tensor(env_ids, device=self.device).to(torch.int64).contiguous()
        self._keep_ids = ids
        return C.c_void_p(ids.data_ptr()), int(ids.numel())

    # ------------------------------------------------------------------ pipeline (one C-ABI call each)
    def init_state(self):
        check(self.lib.dexsim_init_state(self.h, self._stream()), "init_state")

    def process_actions(self, actions, zero_targets=False):
        check(self.lib.dexsim_process_actions(self.h, self._actions_ptr(actions), int(zero_targets), self._stream()), "process_actions")

    def begin_step(self):
        """Open a control step whose action stage ran on the host (what dexsim_process_actions does for the device-side
        step bookkeeping: new reset-gate stamp, contact statistics words)."""
        check(self.lib.dexsim_begin_step(self.h, self._stream()), "begin_step")

    def physics_step(self, gate_on_reset=False):
        check(self.lib.dexsim_physics_step(self.h, int(gate_on_reset), self._stream()), "physics_step")

    def post_physics(self, obs_only=False):
        check(self.lib.dexsim_post_physics(self.h, int(obs_only), self._stream()), "post_physics")

    def step(self, actions):
        check(self.lib.dexsim_step(self.h, self._actions_ptr(actions), self._stream()), "step")

    def reset(self):
        check(self.lib.dexsim_reset(self.h, self._stream()), "reset")

    def reset_idx(self, env_ids):
        ptr, k = self._ids_ptr(env_ids)
        if k:
            check(self.lib.dexsim_reset_idx(self.h, ptr, k, self._stream()), "reset_idx")

    def refresh_body_states(self):
        check(self.lib.dexsim_refresh_body_states(self.h, self._stream()), "refresh_body_states")

    def set_dof_state_indexed(self, env_ids):
        ptr, k = self._ids_ptr(env_ids)
        check(self.lib.dexsim_set_dof_state_indexed(self.h, ptr, k, self._stream()), "set_dof_state_indexed")

    def set_root_state_indexed(self, env_ids):
        ptr, k = self._ids_ptr(env_ids)
        check(self.lib.dexsim_set_root_state_indexed(self.h, ptr, k, self._stream()), "set_root_state_indexed")

    def run_stage(self, stage):
        check(self.lib.dexsim_run_stage(self.h, int(stage), self._stream()), "run_stage")

    def set_step_sink(self, obs=None, rew=None, done=None):
        """Second destination of the next steps' outputs (rows of a rollout buffer): (N, O) f32, (N,) f32, (N,) u8 tensors
        on this device, contiguous; None switches a sink off.  The caller keeps the tensors alive."""
        def ptr(t, dtype, shape):
            if t is None:
                return None
            assert t.is_contiguous() and t.dtype == dtype and tuple(t.shape) == shape and t.device == self.device
            return C.c_void_p(t.data_ptr())
        self._sink_obs = obs                 # (env.py patches a custom pre-action rule's output into the sink row)
        check(self.lib.dexsim_set_step_sink(self.h, ptr(obs, torch.float32, (self.N, int(self.cfg.num_obs))),
                                            ptr(rew, torch.float32, (self.N,)), ptr(done, torch.uint8, (self.N,))), "set_step_sink")

    def set_obs_dict_mode(self, policy_only):
        """False: every obs_dict row is materialised each step (default); True: only obs_buf (the policy keys) is written."""
        check(self.lib.dexsim_set_obs_dict_mode(self.h, 1 if policy_only else 0), "set_obs_dict_mode")

    def set_phase_probe(self, on=True):
        """Switch the phase probe on (returns the zeroed (num_workgroups, 4) int32 tensor the kernels accumulate shader-clock
        ticks into: [:, 0] contact rows + sweeps, [:, 1] whole physics launch, [:, 2] general-path sub-steps, [:, 3] sweeps) or off."""
        self._probe = torch.zeros(self.NS // 64, 4, dtype=torch.int32, device=self.device) if on else None
        check(self.lib.dexsim_set_phase_probe(self.h, None if self._probe is None else C.c_void_p(self._probe.data_ptr())), "set_phase_probe")
        return self._probe

    def set_stats_sink(self, dst):
        """(STAT_WORDS,) f32 row that receives a copy of every step's statistics block (dexsim_set_stats_sink), or None."""
        if dst is not None:
            assert dst.is_contiguous() and dst.dtype == torch.float32 and dst.device == self.device and dst.numel() >= _abi.STAT_USED
        self._sink_stats = dst
        check(self.lib.dexsim_set_stats_sink(self.h, None if dst is None else C.c_void_p(dst.data_ptr())), "set_stats_sink")

    def set_action_copy(self, dst):
        """(N, num_actions) f32 tensor that receives a copy of every step's actions (DexHandBase.actions), or None."""
        if dst is not None:
            assert dst.is_contiguous() and dst.dtype == torch.float32 and dst.device == self.device
            assert tuple(dst.shape) == (self.N, int(self.cfg.num_actions))
        self._actions_copy = dst
        check(self.lib.dexsim_set_action_copy(self.h, None if dst is None else C.c_void_p(dst.data_ptr())), "set_action_copy")

    def step_timing(self, enable):
        """Start (True) / stop (False) the in-situ hipEvent timing of dexsim_step's main launch; stop returns (mean_us, n)."""
        us, n = C.c_float(0.0), C.c_int(0)
        check(self.lib.dexsim_step_timing(self.h, 1 if enable else 0, C.byref(us), C.byref(n)), "step_timing")
        return float(us.value), int(n.value)

    def time_stage(self, stage, launches):
        us = C.c_float(0)
        check(self.lib.dexsim_time_stage(self.h, int(stage), int(launches), self._stream(), C.byref(us)), "time_stage")
        return float(us.value)
