"""Rollout buffer + the only collective on the env path: the end-of-rollout gather of obs / reward / done
into the PPO buffer (BASELINE.json north_star).  One process per GPU; environments are
sharded by rank with no exchange inside step(); `torch.distributed` backend "nccl" is RCCL on ROCm, "gloo"
is used by the CPU tests.

xGMI is point-to-point (7 links/GPU), so the gather is issued as ONE all_gather_into_tensor per tensor per
rollout (41.7 MB/rank at N=4096, T=16, O=158) rather than per step: few, large messages."""
import torch
import torch.distributed as dist


class RolloutBuffer:
    def __init__(self, horizon, num_envs, num_obs, device):
        self.T, self.N, self.O = int(horizon), int(num_envs), int(num_obs)
        self.device = torch.device(device)
        self.obs = torch.zeros(self.T, self.N, self.O, device=self.device)
        self.rew = torch.zeros(self.T, self.N, device=self.device)
        self.done = torch.zeros(self.T, self.N, dtype=torch.uint8, device=self.device)
        self.t = 0
        self._g = None

    def full(self):
        return self.t == self.T

    def add(self, obs, rew, done):
        """Copy one step's outputs (the env returns views of its own buffers, dexhand_base.py:942)."""
        if self.t >= self.T:
            raise RuntimeError("rollout buffer full: call gather()/clear() first")
        self.obs[self.t].copy_(obs)
        self.rew[self.t].copy_(rew)
        self.done[self.t].copy_(done)
        self.t += 1

    def clear(self):
        self.t = 0

    def gather(self):
        """-> (obs (T, R*N, O), rew (T, R*N), done (T, R*N)) on every rank; env index = rank*N + local."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            self.t = 0
            return self.obs, self.rew, self.done
        R = dist.get_world_size()
        if self._g is None:
            # rank-major concatenation along dim 0 (the layout both RCCL and gloo accept)
            self._g = (torch.empty(R * self.T, self.N, self.O, device=self.device),
                       torch.empty(R * self.T, self.N, device=self.device),
                       torch.empty(R * self.T, self.N, dtype=torch.uint8, device=self.device))
        T, N, O = self.T, self.N, self.O
        dist.all_gather_into_tensor(self._g[0], self.obs)
        dist.all_gather_into_tensor(self._g[1], self.rew)
        dist.all_gather_into_tensor(self._g[2], self.done)
        go, gr, gd = self._g[0].view(R, T, N, O), self._g[1].view(R, T, N), self._g[2].view(R, T, N)
        self.t = 0
        return (go.permute(1, 0, 2, 3).reshape(T, R * N, O), gr.permute(1, 0, 2).reshape(T, R * N),
                gd.permute(1, 0, 2).reshape(T, R * N))
