"""Adapter that drives the HIP kernels through the C-ABI with the same stage interface as oracle.Oracle,
so that the golden replays and the oracle-parity tests run unchanged on the GPU."""
import numpy as np
import torch

from dexrobot_isaac_amd import _abi
from dexrobot_isaac_amd.core import DexSimCore



class HipBackend:
    def __init__(self, sim_cfg, model_struct, device="cuda:0", fused=True):
        self.fused = fused        # True: production k_substep; False: stand-alone k_dynamics + k_solve
        self.core = DexSimCore(sim_cfg, model_struct, device)
        self.n = self.core.N
        self._posted = False

    # -- state exchange (SoA [rows][N] float64, like the oracle wrapper)
    def set(self, name, value):
        f = self.core.field(name)
        v = np.broadcast_to(np.asarray(value, dtype=np.float64), tuple(f.shape))
        f.copy_(torch.as_tensor(np.array(v, copy=True)).to(f.dtype))

    def get(self, name):
        return self.core.field(name).detach().cpu().numpy().astype(np.float64)

    def warm_cache(self):
        """The contact solver's warm-start cache in the oracle's terms: (wlam [88 x 3][N], wtag [88][N], wgen [N]).  The arena keeps
        one float4 per cache slot and env, [slot][env][4] = impulses + the tag's integer bits."""
        q = self.core.field("wlam").detach().cpu().numpy().reshape(-1).reshape(_abi.NWKEY, -1, 4)
        lam = np.ascontiguousarray(q[:, :, :3].transpose(0, 2, 1)).reshape(_abi.NWKEY * 3, -1).astype(np.float64)
        tag = np.ascontiguousarray(q[:, :, 3]).view(np.int32).astype(np.int64)
        return lam, tag, self.core.field("wgen").detach().cpu().numpy()[0].astype(np.int64)

    def obs_buf(self):
        return self.core.obs_buf.detach().cpu().numpy()

    def stats(self):
        return self.core.stats.detach().cpu().numpy()

    def set_reset_samples(self, s):
        self.core.set_reset_samples(s)

    def contacts(self, env):
        k = int(self.get("ncontact")[0, env])
        # cgeom is a quad-layout field: list entry kk = quads 2 kk, 2 kk + 1 of [quad][env][4] (p.xyz n.x | n.yz gap mu)
        off, rows, _ = self.core.fields["cgeom"]
        quads = self.core.arena[off: off + rows * self.core.NS].view(rows // 4, self.core.NS, 4)
        g = quads[:, env].detach().cpu().numpy().astype(np.float64).reshape(_abi.KMAX, 8)[:k]
        code = self.get("ccode")[:, env][:k].astype(int)
        out = np.zeros((k, 10))
        out[:, :8] = g
        out[:, 8] = code & 3
        out[:, 9] = (code >> 2) & 31
        return out

    # -- pipeline stages
    def process_actions(self, actions, zero_targets=False):
        self.core.process_actions(torch.as_tensor(np.ascontiguousarray(actions), dtype=torch.float32,
                                                  device=self.core.device), zero_targets)

    def compute_observations(self):
        self.core.run_stage(_abi.STAGE["POST"] + 100)

    def l2_step_no_reset(self):
        self.core.run_stage(_abi.STAGE["POST"])
        self.core.run_stage(_abi.STAGE["FINALIZE"])     # step statistics; opens the next staged control step

    def reset_flagged_no_physics(self):
        self.core.run_stage(_abi.STAGE["RESET"])        # flagged envs, no device-side gate, no physics

    def substep(self, last=True):
        if self.fused:
            self.core.run_stage(_abi.STAGE["SUBSTEP"])
        else:
            self.core.run_stage(_abi.STAGE["DYNAMICS"])
            self.core.run_stage(_abi.STAGE["SOLVE"])

    def publish(self):
        self.core.run_stage(_abi.STAGE["PUBLISH"])

    def physics_step(self):
        self.core.physics_step(False)

    def step(self, actions):
        self.core.step(torch.as_tensor(np.ascontiguousarray(actions), dtype=torch.float32, device=self.core.device))
        torch.cuda.synchronize()
        return (self.obs_buf(), self.core.rew_buf.cpu().numpy(), self.core.reset_buf.cpu().numpy())

    def reset(self):
        self.core.reset()
        torch.cuda.synchronize()
        return self.obs_buf()

    def reset_idx(self, ids):
        self.core.reset_idx(torch.as_tensor(np.asarray(ids, dtype=np.int64)))
