"""Rollout buffer + the only collectives on the env path: the end-of-rollout gather of obs / reward / done
into the PPO buffer (BASELINE.json north_star) and the reduction of the whole-population
statistics.  One process per GPU; environments are sharded by rank with no exchange inside step();
`torch.distributed` backend "nccl" is RCCL on ROCm, "gloo" is used by the CPU tests.

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

Whole-population statistics (round 3).  TerminationManager's logging means / rates and `consecutive_successes`
(reference components/termination/termination_manager.py:160-185, 323-339) are statistics over ALL envs.  With the envs
sharded, each rank's kernels compute them over its own shard; the step's closing launch also drops its statistics block
into row t of the slot (`dexsim_set_stats_sink`), and `gather_async()` adds ONE all-reduce (T x 20 floats, SUM) per rollout:
means and rates are averaged over the ranks (equal shards), counts are summed, and the global consecutive-successes
counter is re-derived step by step from the global "some env succeeded in this step" flag.  No collective inside step().
"""
import torch
import torch.distributed as dist

from . import _abi


class _Slot:
    def __init__(self, T, N, O, device):
        self.obs = torch.zeros(T, N, O, device=device)
        self.rew = torch.zeros(T, N, device=device)
        self.done = torch.zeros(T, N, dtype=torch.uint8, device=device)
        self.stats = torch.zeros(T, _abi.STAT_WORDS, device=device)      # one statistics block per step (this rank's shard)
        self.stats_sum = None                                            # ... summed over the ranks (gather_async)
        self.gathered = None
        self.work = []


class RolloutHandle:
    """What `gather_async()` returns.  Calling it waits for the rollout's collectives and returns the full-batch views
    (obs (T, R*N, O), rew (T, R*N), done (T, R*N)) on the receiving rank(s), None on the others; `stats()` returns the
    whole-population statistics of the rollout's T steps on every rank."""

    def __init__(self, buf, slot):
        self._buf, self._slot = buf, slot
        self._stats = None

    def __call__(self):
        self._buf._wait(self._slot)
        return self._buf._views(self._slot)

    def stats(self):
        if self._stats is None:
            self._buf._wait(self._slot)
            # the global consecutive-successes counter is carried from rollout to rollout: read the handles' stats in
            # rollout order (one handle per rollout; each computes its statistics once and caches them)
            self._stats = self._buf._reduced_stats(self._slot)
        return self._stats


class RolloutBuffer:
    def __init__(self, horizon, num_envs, num_obs, device, slots=2, mode="learner", learner_rank=0,
                 max_consecutive_successes=50):
        if mode not in ("learner", "all"):
            raise ValueError("mode must be 'learner' (gather to one rank) or 'all' (all-gather)")
        self.mode, self.learner_rank = mode, int(learner_rank)
        self.T, self.N, self.O = int(horizon), int(num_envs), int(num_obs)
        self.device = torch.device(device)
        self.slots = [_Slot(self.T, self.N, self.O, self.device) for _ in range(max(1, int(slots)))]
        self.cur = 0
        self.t = 0
        # the GLOBAL consecutive-successes counter (termination_manager.py:323-339), carried from rollout to rollout by
        # RolloutHandle.stats() (read the handles' statistics in rollout order)
        self.max_consecutive_successes = int(max_consecutive_successes)
        self.consecutive_successes = 0

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

    def add(self, obs, rew, done, stats=None):
        """Copy one step's outputs (the env returns views of its own buffers, dexhand_base.py:942); `stats` = the env
        core's statistics block of the step (optional)."""
        if self.t >= self.T:
            raise RuntimeError("rollout buffer full: call gather()/gather_async()/clear() first")
        s = self.slots[self.cur]
        if self.t == 0:
            self._wait(s)                      # the slot may still be the source of an in-flight gather
        s.obs[self.t].copy_(obs)
        s.rew[self.t].copy_(rew)
        s.done[self.t].copy_(done)
        if stats is not None:
            s.stats[self.t, : stats.numel()].copy_(stats)
        self.t += 1

    def sink(self, core):
        """Copy-free variant of add(): point the simulator's step sink (dexsim_set_step_sink, dexsim_set_stats_sink) at row
        t of the slot being filled and advance; call BEFORE core.step().  The step's flush then writes obs / rew / done and
        the step's closing launch its statistics block straight into the rollout buffer."""
        if self.t >= self.T:
            raise RuntimeError("rollout buffer full: call gather()/gather_async()/clear() first")
        s = self.slots[self.cur]
        if self.t == 0:
            self._wait(s)
        core.set_step_sink(s.obs[self.t], s.rew[self.t], s.done[self.t])
        if hasattr(core, "set_stats_sink"):
            core.set_stats_sink(s.stats[self.t])
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

    def _reduced_stats(self, slot):
        """Whole-population statistics of the slot's T steps from the rank-summed statistics blocks."""
        S = _abi.STAT
        R = dist.get_world_size() if self._distributed() else 1
        tot = (slot.stats_sum if slot.stats_sum is not None else slot.stats).detach().to("cpu", torch.float64)
        mean = tot / R                                               # equal shards: mean of the ranks' means
        out = {"success_rate": mean[:, S["SUCCESS_RATE"]], "failure_rate": mean[:, S["FAILURE_RATE"]],
               "timeout_rate": mean[:, S["TIMEOUT_RATE"]], "num_resets": tot[:, S["NUM_RESETS"]],
               "mean_contacts": mean[:, S["MEAN_CONTACTS"]], "mean_hand_contacts": mean[:, S["MEAN_HAND_CONTACTS"]],
               "physics_steps": mean[:, S["PHYSICS_STEPS"]]}
        for i, name in enumerate(_abi.SUCCESS_CRITERIA):
            out[f"success_{name}"] = mean[:, S["SUCC_MEAN"] + i]
        for i, name in enumerate(_abi.FAILURE_CRITERIA):
            out[f"failure_{name}"] = mean[:, S["FAIL_MEAN"] + i]
        # update_consecutive_successes (termination_manager.py:323-339) over the global population: +1 in a step in which
        # ANY env of ANY rank terminated with success, else back to 0, capped
        cs, seq = int(self.consecutive_successes), []
        for t in range(self.T):
            cs = min(cs + 1, self.max_consecutive_successes) if float(tot[t, S["SUCCESS_RATE"]]) > 0.0 else 0
            seq.append(cs)
        out["consecutive_successes"] = torch.tensor(seq, dtype=torch.float64)
        self.consecutive_successes = cs
        return out

    def gather_async(self):
        """Enqueue the gather of the slot just filled (+ the all-reduce of its statistics rows) and switch to the next slot.
        Returns a RolloutHandle: `handle()` -> (obs (T, R*N, O), rew (T, R*N), done (T, R*N)), env index = rank*N + local,
        on the receiving rank(s) (None on the other ranks of mode "learner", once their sends are done); `handle.stats()`
        -> the whole-population statistics of the T steps, on every rank."""
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
            if slot.stats_sum is None:
                slot.stats_sum = torch.empty_like(slot.stats)
            slot.stats_sum.copy_(slot.stats)                          # (the slot's own rows stay this rank's)
            slot.work.append(dist.all_reduce(slot.stats_sum, op=dist.ReduceOp.SUM, async_op=True))
        self.cur = (self.cur + 1) % len(self.slots)
        self.t = 0
        return RolloutHandle(self, slot)

    def gather(self):
        """Blocking form: gather the slot just filled and return the full-batch views."""
        return self.gather_async()()
