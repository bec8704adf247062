"""Rollout buffer + the only collective on the env path: the end-of-rollout gather of obs / reward / done
into the PPO buffer (BASELINE.json north_star).  One process per GPU; environments are
sharded by rank with no exchange inside step(); `torch.distributed` backend "nccl" is RCCL on ROCm, "gloo"
is used by the CPU tests.

xGMI is point-to-point (7 links/GPU), so the gather is issued as ONE all_gather_into_tensor per tensor per
rollout (41.7 MB/rank at N=4096, T=16, O=158) rather than per step: few, large messages.  The buffer is
double-buffered: `gather_async()` enqueues the collectives (RCCL runs them on its own HIP stream) and the
env keeps stepping into the other slot, so the ~1-3 ms of xGMI traffic overlaps the next rollout's ~3.5 ms of
simulation instead of serialising with it.
"""
import torch
import torch.distributed as dist


class _Slot:
    def __init__(self, T, N, O, device):
        self.obs = torch.zeros(T, N, O, device=device)
        self.rew = torch.zeros(T, N, device=device)
        self.done = torch.zeros(T, N, dtype=torch.uint8, device=device)
        self.gathered = None
        self.work = []


class RolloutBuffer:
    def __init__(self, horizon, num_envs, num_obs, device, slots=2):
        self.T, self.N, self.O = int(horizon), int(num_envs), int(num_obs)
        self.device = torch.device(device)
        self.slots = [_Slot(self.T, self.N, self.O, self.device) for _ in range(max(1, int(slots)))]
        self.cur = 0
        self.t = 0

    # views of the slot being filled (kept for callers that index the buffer directly)
    @property
    def obs(self):
        return self.slots[self.cur].obs

    @property
    def rew(self):
        return self.slots[self.cur].rew

    @property
    def done(self):
        return self.slots[self.cur].done

    def full(self):
        return self.t == self.T

    def add(self, obs, rew, done):
        """Copy one step's outputs (the env returns views of its own buffers, dexhand_base.py:942)."""
        if self.t >= self.T:
            raise RuntimeError("rollout buffer full: call gather()/gather_async()/clear() first")
        s = self.slots[self.cur]
        if self.t == 0:
            self._wait(s)                      # the slot may still be the source of an in-flight gather
        s.obs[self.t].copy_(obs)
        s.rew[self.t].copy_(rew)
        s.done[self.t].copy_(done)
        self.t += 1

    def sink(self, core):
        """Copy-free variant of add(): point the simulator's step sink (dexsim_set_step_sink) at row t of the slot being
        filled and advance; call BEFORE core.step().  The step's flush then writes obs / rew / done straight into the
        rollout buffer."""
        if self.t >= self.T:
            raise RuntimeError("rollout buffer full: call gather()/gather_async()/clear() first")
        s = self.slots[self.cur]
        if self.t == 0:
            self._wait(s)
        core.set_step_sink(s.obs[self.t], s.rew[self.t], s.done[self.t])
        self.t += 1

    def clear(self):
        self.t = 0

    @staticmethod
    def _distributed():
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    @staticmethod
    def _wait(slot):
        for w in slot.work:
            w.wait()                           # stream-level wait on the device; does not block the host for nccl
        slot.work = []

    def _views(self, slot):
        if not self._distributed():
            return slot.obs, slot.rew, slot.done
        R, T, N, O = dist.get_world_size(), self.T, self.N, self.O
        go, gr, gd = slot.gathered
        return (go.view(R, T, N, O).permute(1, 0, 2, 3).reshape(T, R * N, O),
                gr.view(R, T, N).permute(1, 0, 2).reshape(T, R * N), gd.view(R, T, N).permute(1, 0, 2).reshape(T, R * N))

    def gather_async(self):
        """Enqueue the gather of the slot just filled and switch to the next slot.  Returns a handle; call
        `handle()` to obtain (obs (T, R*N, O), rew (T, R*N), done (T, R*N)) -- env index = rank*N + local."""
        slot = self.slots[self.cur]
        if self._distributed():
            R = dist.get_world_size()
            if slot.gathered is None:
                # rank-major concatenation along dim 0 (the layout both RCCL and gloo accept)
                slot.gathered = (torch.empty(R * self.T, self.N, self.O, device=self.device),
                                 torch.empty(R * self.T, self.N, device=self.device),
                                 torch.empty(R * self.T, self.N, dtype=torch.uint8, device=self.device))
            slot.work = [dist.all_gather_into_tensor(slot.gathered[0], slot.obs, async_op=True),
                         dist.all_gather_into_tensor(slot.gathered[1], slot.rew, async_op=True),
                         dist.all_gather_into_tensor(slot.gathered[2], slot.done, async_op=True)]
        self.cur = (self.cur + 1) % len(self.slots)
        self.t = 0

        def handle():
            self._wait(slot)
            return self._views(slot)
        return handle

    def gather(self):
        """Blocking form: gather the slot just filled and return the full-batch views."""
        return self.gather_async()()
