"""Rollout buffer + the only collective on the env path: the end-of-rollout gather of obs / reward / done
into the PPO buffer (BASELINE.json north_star).  One process per GPU; environments are
sharded by rank with no exchange inside step(); `torch.distributed` backend "nccl" is RCCL on ROCm, "gloo"
is used by the CPU tests.

xGMI is point-to-point (7 links/GPU), so the gather is issued as ONE collective per tensor per rollout
(41.7 MB/rank at N=4096, T=16, O=158) rather than per step: few, large messages.  Two modes:

  mode="learner" (default): `gather` to one learner rank (north_star: "gather into the PPO buffer").  Every other rank
      only SENDS its 41.7 MB over its direct xGMI link to the learner; the learner receives (R-1) x 41.7 MB spread over
      its 7 links.  This is 1/R of the receive traffic of an all-gather on the non-learner ranks.
  mode="all": `all_gather_into_tensor` -- every rank ends up with the full batch (data-parallel learners that each
      train on the whole buffer).  R x the receive traffic; at ~0.1 ms per control step a 16-step rollout is only
      ~1.6 ms of simulation, which an 8-rank all-gather (8 x 41.7 MB into every GPU) no longer hides.

The buffer is double-buffered: `gather_async()` enqueues the collectives (RCCL runs them on its own HIP stream) and
the env keeps stepping into the other slot, so the xGMI traffic overlaps the next rollout's simulation instead of
serialising with it; bench.py prints the simulation ms per rollout next to the collective ms so that the overlap (or
its absence) is visible.
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
    def __init__(self, horizon, num_envs, num_obs, device, slots=2, mode="learner", learner_rank=0):
        if mode not in ("learner", "all"):
            raise ValueError("mode must be 'learner' (gather to one rank) or 'all' (all-gather)")
        self.mode, self.learner_rank = mode, int(learner_rank)
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

    def _is_receiver(self):
        return self.mode == "all" or dist.get_rank() == self.learner_rank

    def _views(self, slot):
        if not self._distributed():
            return slot.obs, slot.rew, slot.done
        if not self._is_receiver():
            return None                        # mode "learner": only the learner rank holds the full batch
        R, T, N, O = dist.get_world_size(), self.T, self.N, self.O
        go, gr, gd = slot.gathered
        return (go.view(R, T, N, O).permute(1, 0, 2, 3).reshape(T, R * N, O),
                gr.view(R, T, N).permute(1, 0, 2).reshape(T, R * N), gd.view(R, T, N).permute(1, 0, 2).reshape(T, R * N))

    def gather_async(self):
        """Enqueue the gather of the slot just filled and switch to the next slot.  Returns a handle; call
        `handle()` to obtain (obs (T, R*N, O), rew (T, R*N), done (T, R*N)) -- env index = rank*N + local -- on the
        receiving rank(s); on the other ranks of mode "learner" the handle returns None once their sends are done."""
        slot = self.slots[self.cur]
        if self._distributed():
            R = dist.get_world_size()
            if slot.gathered is None and self._is_receiver():
                # rank-major concatenation along dim 0 (the layout both RCCL and gloo accept)
                slot.gathered = (torch.empty(R * self.T, self.N, self.O, device=self.device),
                                 torch.empty(R * self.T, self.N, device=self.device),
                                 torch.empty(R * self.T, self.N, dtype=torch.uint8, device=self.device))
            src = (slot.obs, slot.rew, slot.done)
            if self.mode == "all":
                slot.work = [dist.all_gather_into_tensor(g, t, async_op=True) for g, t in zip(slot.gathered, src)]
            else:
                recv = dist.get_rank() == self.learner_rank
                slot.work = [dist.gather(t, list(slot.gathered[i].chunk(R, dim=0)) if recv else None,
                                         dst=self.learner_rank, async_op=True) for i, t in enumerate(src)]
        self.cur = (self.cur + 1) % len(self.slots)
        self.t = 0

        def handle():
            self._wait(slot)
            return self._views(slot)
        return handle

    def gather(self):
        """Blocking form: gather the slot just filled and return the full-batch views."""
        return self.gather_async()()
