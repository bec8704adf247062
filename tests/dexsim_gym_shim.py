"""Level-2 binding of INTEGRATION.md as a real, tested file: the subset of `isaacgym.gymapi`'s gym object that the
reference's step / reset path touches, forwarded to libdexsim through ctypes.

A maintainer of the reference would drop this file next to `dexhand_env/components/physics/physics_manager.py` and hand
an instance to `PhysicsManager` / `TensorManager` / `ActionProcessor` / `ResetManager` instead of
`gymapi.acquire_gym()`.  The reference's call sites each method stands in for are cited per method.  Allocation of the
device tensors is left to `dexrobot_isaac_amd.core.DexSimCore` here (it is plumbing: torch.zeros + dexsim_bind); every
simulation call below goes straight to the C-ABI.

Exercised by tests/test_gpu_parity.py::test_level2_gym_shim_matches_fused_step.
"""
import ctypes as C

import torch

from dexrobot_isaac_amd.core import DexSimCore


class DexSimGym:
    def __init__(self, sim_cfg, model_struct, device="cuda:0"):
        self.core = DexSimCore(sim_cfg, model_struct, device)       # dexsim_create + dexsim_bind + dexsim_init_state
        self.lib, self.h = self.core.lib, self.core.h
        self.actors_per_env = self.core.num_actors
        # gymtorch.wrap_tensor(gym.acquire_*_tensor(sim)) (tensor_manager.py:173,253,282,311): the tensors bound to the library
        self.dof_state = self.core.dof_state                        # (N, 26, 2)
        self.actor_root_state = self.core.root_state                # (N, A, 13)
        self.rigid_body_state = self.core.rigid_body_states         # (N, B, 13)
        self.net_contact_force = self.core.contact_forces_all       # (N, B, 3)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.core.device).cuda_stream)

    def _ok(self, rc):
        if rc != 0:
            raise RuntimeError(self.lib.dexsim_last_error().decode())

    # ---- physics_manager.py:92-109: gym.simulate + gym.fetch_results + the four refreshes
    def simulate(self, sim=None):
        self._ok(self.lib.dexsim_physics_step(self.h, 0, self._stream()))       # one launch for a whole sim.dt

    def fetch_results(self, sim=None, wait=True):
        pass                                                                    # stream-ordered: no blocking wait

    def refresh_dof_state_tensor(self, sim=None):
        pass                                                                    # published by dexsim_physics_step

    def refresh_actor_root_state_tensor(self, sim=None):
        pass

    def refresh_rigid_body_state_tensor(self, sim=None):
        self._ok(self.lib.dexsim_refresh_body_states(self.h, self._stream()))   # (N, B, 13) and (N, B, 3), on demand

    def refresh_net_contact_force_tensor(self, sim=None):
        pass                                                                    # same call as above

    # ---- action_processor.py:348: gym.set_dof_position_target_tensor(sim, unwrap(full_dof_targets))
    def set_dof_position_target_tensor(self, sim, targets):
        self.core.field("targets").copy_(targets.t())                           # (N, 26) AoS -> arena rows [26][N]
        self.core.full_dof_targets.copy_(targets)

    # ---- physics_manager.py:146-151, reset_manager.py:153-158 (indices are GLOBAL actor indices, int32)
    def set_dof_state_tensor_indexed(self, sim, state, actor_indices, count):
        envs = (actor_indices.long() // self.actors_per_env)[:count].contiguous()
        self._ok(self.lib.dexsim_set_dof_state_indexed(self.h, C.c_void_p(envs.data_ptr()), int(envs.numel()), self._stream()))
        self._keep = envs

    def set_actor_root_state_tensor_indexed(self, sim, root, actor_indices, count):
        envs = torch.unique(actor_indices.long()[:count] // self.actors_per_env).contiguous()
        self._ok(self.lib.dexsim_set_root_state_indexed(self.h, C.c_void_p(envs.data_ptr()), int(envs.numel()), self._stream()))
        self._keep = envs
