#!/usr/bin/env python
"""bench.py -- env-steps/sec of random-action BlindGrasping at num_envs=4096 per GPU (BASELINE.json metric,
configs[2] at N=1; configs[3] = the same sharded over N GPUs, weak scaling).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one env.step() for every env of the rank: action processing, one physics step (4 sub-steps of
dynamics + PGS contact solve), fused obs/reward/termination, in-step resets and the reference's conditional
extra physics step -- nothing is skipped.  Synthetic data: random actions 2*U(0,1)-1 (the law of the
reference's random-action mode, dexhand_base.py:856), pre-generated in HBM before the timed region.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def cpu_baseline(sim_cfg_factory, seconds_budget=20.0):
    """The CPU oracle (scalar restatement of the same algorithm; 'port', not PhysX) on the host cores,
    bounded sample of the same workload."""
    import numpy as np
    from oracle.oracle import Oracle
    n = 1024
    sc, model = sim_cfg_factory(n)
    best = None
    avail = len(os.sched_getaffinity(0))
    for cores in sorted({4, min(avail, 16)}):
        o = Oracle(sc, model.to_struct(), threads=cores)
        o.reset()
        rng = np.random.default_rng(1234)
        acts = [(2 * rng.random((n, 18)) - 1).astype(np.float32) for _ in range(4)]
        o.step(acts[0])
        t0, k = time.time(), 0
        while time.time() - t0 < seconds_budget / 2 and k < 40:
            o.step(acts[k % 4])
            k += 1
        v = n * k / (time.time() - t0)
        if best is None or v > best["value"]:
            best = {"value": v, "unit": "env-steps/s", "cores": cores, "kind": "port",
                    "sample": f"BlindGrasping N={n}, {k} control steps, random actions, oracle/dexsim_oracle.c with OpenMP"}
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--num-envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--horizon", type=int, default=16, help="rollout length between RCCL gathers (gpus > 1)")
    ap.add_argument("--dr", action="store_true", help="BASELINE config #5: per-env box mass/friction randomisation")
    ap.add_argument("--task", default="BlindGrasping", choices=["BlindGrasping", "BaseTask"],
                    help="BaseTask + --control-mode position + --num-envs 1024 = BASELINE configs[1] (articulated FD only, no object)")
    ap.add_argument("--control-mode", default=None, choices=["position", "position_delta"])
    ap.add_argument("--rollout", action="store_true", help="fill the rollout buffer also when --gpus 1 (the multi-GPU code path minus the collective)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stagger", action="store_true", help="skip the secondary staggered-episode measurement")
    args = ap.parse_args()

    import torch.distributed as dist
    from dexrobot_isaac_amd import _abi
    from dexrobot_isaac_amd.config import build_sim_config, default_cfg
    from dexrobot_isaac_amd.core import DexSimCore
    from dexrobot_isaac_amd.rollout import RolloutBuffer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    def factory(n):
        cfg = default_cfg(args.task)
        cfg["env"]["numEnvs"] = n
        if args.control_mode:
            cfg["task"]["controlMode"] = args.control_mode
        dr = {"mass": (0.05, 0.2), "friction": (0.5, 1.5), "seed": 4242 + rank} if args.dr else None
        return build_sim_config(cfg, dr=dr)

    N = args.num_envs
    sc, model = factory(N)
    sc.seed = 42 + rank                      # rank-local reset stream
    core = DexSimCore(sc, model.to_struct(), device)
    core.reset()
    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + rank)
    n_act = 64                               # distinct pre-generated action batches, cycled
    actions = 2.0 * torch.rand(n_act, N, 18, device=device, generator=gen) - 1.0
    rollout = RolloutBuffer(args.horizon, N, sc.num_obs, device) if (world > 1 or args.rollout) else None

    pending = []

    def run(k):
        for i in range(k):
            if rollout is not None:
                rollout.sink(core)           # this step's obs / rew / done land in the rollout slot: no copy kernels
            core.step(actions[i % n_act])
            if rollout is not None:
                if rollout.full():
                    # RCCL all-gather over xGMI, once per rollout, overlapped with the next rollout's simulation
                    pending.append(rollout.gather_async())
                    if len(pending) > 1:
                        pending.pop(0)()     # the previous rollout's gather must have landed by now
        while pending:
            pending.pop(0)()

    run(args.warmup)
    resets0 = float(core.field("reset_count").sum().item())
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()                             # HIP events on the launch stream, around the whole timed region
    run(args.steps)
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    resets = float(core.field("reset_count").sum().item()) - resets0

    if rank == 0:
        total_envs = N * world
        value = total_envs * args.steps / dt
        # live per-kernel timing with HIP events on the launch stream (one launch = one sub-step of all envs)
        kbar = float(core.stats[_abi.STAT["MEAN_CONTACTS"]].item())
        # The production launch (k_physics4<false>: actions + 4 sub-steps + post-physics) is issued once per control step,
        # so its mean duration over the timed region = GPU time between the two HIP events / steps, minus nothing: the
        # figure therefore also contains the device-gated no-op launch (~3 us) and the dispatch gaps -- an upper bound.
        # A hipEvent pair around each single launch is NOT used: the event's release/acquire fences make the kernel
        # start from a cold L2 (rocprofv3 shows 125 us instead of 94 us for such launches, profiles/round1_d_*).
        t_step = ev0.elapsed_time(ev1) * 1e3 / args.steps
        t_sub = core.time_stage(_abi.STAGE["SUBSTEP"], 50)     # a single sub-step (dynamics + solve + integrate + publish) as its own launch
        t_solve = core.time_stage(_abi.STAGE["SOLVE"], 50)     # the same sweeps as a stand-alone kernel
        t_dyn = core.time_stage(_abi.STAGE["DYNAMICS"], 50)
        t_post = core.time_stage(_abi.STAGE["POST"], 20)
        t_pub = core.time_stage(_abi.STAGE["PUBLISH"], 20)
        peak = 8000.0
        step_bytes = 2560.0                            # algorithmic bytes of a whole env-step
        # algorithmic HBM bytes per env per launch
        b_sub = (632.0 + 204.0) * N                    # q, qd, targets, box in; q, qd, box out; cforce (last sub-step)
        b_solve = (256.0 + 60.0 * kbar) * N
        b_dyn = (500.0 + 36.0 * kbar) * N
        pmc = {}
        try:                                           # HBM traffic from the separate rocprofv3 --pmc passes
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
        except Exception:
            pass

        def roof(kernel, bytes_, us):
            a = bytes_ / (us * 1e-6) / 1e9
            tr = None
            rec = pmc.get(kernel)
            if rec and int(rec.get("num_envs", -1)) == N:
                # gfx950: FETCH_SIZE tallies 64 B per 128-B request -> x2 (MI355X_MICROARCH.md, HBM); counters are in KB
                tr = (2.0 * rec["fetch_kb"] + rec["write_kb"]) * 1024.0
            return {"bound": "hbm", "achieved": a, "peak": peak, "unit": "GB/s", "frac": a / peak, "traffic": tr,
                    "kernel": kernel, "avg_us": us, "algorithmic_bytes": bytes_, "mean_contacts": kbar}

        out = {
            "metric": f"env-steps/sec {args.task} num_envs={N} per MI355X (random actions)",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"{args.task} num_envs={N}/GPU" + (" (BASELINE configs[2]; configs[3] when n_gpus>1)"
                                    if args.task == "BlindGrasping" and N == 4096 and not args.dr else "")),
                       "control_mode": args.control_mode or "task default",
                       "num_envs_per_gpu": N, "sim_dt": 0.01, "substeps": 4, "pgs_iterations": 16,
                       "parallelism": f"env-shard x{world}", "rollout_gather_horizon": args.horizon if world > 1 else None,
                       "domain_randomisation": bool(args.dr)},
            "resets_per_step": resets / args.steps,
            # dominant kernel = the one launch that carries a whole control step (k_physics4<false> with the action and
            # post-physics blocks); algorithmic bytes = SURVEY 8d's whole-env-step figure
            "roofline": roof("k_physics4", step_bytes * N, t_step),
            "roofline_substep": roof("k_substep", b_sub, t_sub),
            "roofline_contact_solve": roof("k_solve", b_solve, t_solve),
            "roofline_dynamics": roof("k_dynamics", b_dyn, t_dyn),
            "roofline_whole_step": {"bound": "hbm", "achieved": step_bytes * value / world / 1e9, "peak": peak,
                                    "unit": "GB/s", "frac": step_bytes * value / world / 1e9 / peak, "traffic": None},
            "kernel_us": {"k_physics4_step": t_step, "k_substep": t_sub, "k_dynamics": t_dyn, "k_solve": t_solve, "k_post": t_post, "k_publish": t_pub},
        }
        if world == 1 and not args.no_stagger:
            # Secondary figure, NOT `value`: the timed region above starts from a fresh env, like the reference's own run,
            # so all episodes time out in the same control step and the conditional second physics step of
            # reset_manager.py:180 runs once per ~300 steps.  In training the episodes de-synchronise and some env
            # resets in (almost) every control step; this run staggers the episode clocks to show that regime.
            # (random actions end every episode through the stage-1 pre-grasp check at t = 4 s = 200 control steps, so the
            # stage clock and the episode clock are staggered together)
            es, tis = core.field("episode_step"), core.field("time_in_stage")
            k = torch.randint(0, 199, es.shape, device=device, generator=gen)
            es.copy_(k.to(es.dtype))
            tis.copy_(k.to(tis.dtype) * float(sc.control_dt))
            rollout = None                   # (the secondary figure is measured without the rollout sink)
            core.set_step_sink(None, None, None)
            run(50)
            r0 = float(core.field("reset_count").sum().item())
            torch.cuda.synchronize()
            ts = time.perf_counter()
            run(200)
            torch.cuda.synchronize()
            ds = time.perf_counter() - ts
            out["staggered_resets"] = {"value": N * 200 / ds, "unit": "env-steps/s", "ms_per_step": ds / 200 * 1e3,
                                       "resets_per_step": (float(core.field("reset_count").sum().item()) - r0) / 200}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(factory)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
