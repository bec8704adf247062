#!/usr/bin/env python
"""bench.py -- env-steps/sec of random-action BlindGrasping at num_envs=4096 per GPU (BASELINE.json metric,
configs[2] at N=1; configs[3] = the same sharded over N GPUs, weak scaling).

    python bench.py --gpus N --steps K --warmup W           (N > 1 without WORLD_SIZE: starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one env.step() for every env of the rank: action processing, one physics step (4 sub-steps of
dynamics + PGS contact solve), fused obs/reward/termination, in-step resets and the reference's conditional
extra physics step -- nothing is skipped.  Synthetic data: random actions 2*U(0,1)-1 (the law of the
reference's random-action mode, dexhand_base.py:856), pre-generated in HBM before the timed region.

Besides the contract line's `value` the JSON carries three secondary regimes that are never `value`:
`staggered_resets` (episode clocks de-synchronised: some env resets in almost every step, as in training),
`contact_rich` (every hand lowered onto its box: the regime a grasping policy lives in, with the contact-solve
roofline measured there; `us_per_physics_step` = round 1's window right after the teleport, `settled` = the 100 steps
after it) and `cpu_baseline` (the CPU oracle on the host cores).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0                      # MI355X_MICROARCH.md: HBM3E ~8 TB/s
STEP_BYTES = 2560.0                        # algorithmic bytes of a whole env-step


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--num-envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--horizon", type=int, default=16, help="rollout length between RCCL gathers (gpus > 1)")
    ap.add_argument("--gather", default="learner", choices=["learner", "all"],
                    help="end-of-rollout collective: gather to the learner rank (default) or all-gather")
    ap.add_argument("--dr", action="store_true", help="BASELINE config #5: per-env box mass/friction randomisation")
    ap.add_argument("--task", default="BlindGrasping", choices=["BlindGrasping", "BaseTask"],
                    help="BaseTask + --control-mode position + --num-envs 1024 = BASELINE configs[1] (articulated FD only, no object)")
    ap.add_argument("--control-mode", default=None, choices=["position", "position_delta"])
    ap.add_argument("--rollout", action="store_true", help="fill the rollout buffer also when --gpus 1 (the multi-GPU code path minus the collective)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stagger", action="store_true", help="skip the secondary staggered-episode measurement")
    ap.add_argument("--no-contact-rich", action="store_true", help="skip the secondary contact-rich measurement")
    ap.add_argument("--no-training-like", action="store_true", help="skip the secondary training-like measurement (make_env().step, contacts + reset churn)")
    ap.add_argument("--preroll", type=int, default=0,
                    help="control steps run on a THROW-AWAY env right before every timed region's synchronisation, to keep the GPU busy "
                         "(stated in the output; measured in round 3: no effect on the 20-step figure, so off by default)")
    ap.add_argument("--obs-dict", default="policy", choices=["policy", "all"],
                    help="policy (default): only obs_buf is written, as in a training loop (env.obsDict = policy); all: every step also "
                         "materialises the 392 rows behind get_observations_dict() (the product's default for API parity)")
    ap.add_argument("--joint-limit-rows", action="store_true", help="sim.dexsim_joint_limit_rows: joint limits as unilateral solver rows (option; see DESIGN.md section 4)")
    ap.add_argument("--substeps", type=int, default=None, help="sim.substeps (default: the task's 4; 32 = cfg/physics/accurate.yaml)")
    ap.add_argument("--iterations", type=int, default=None, help="sim.physx.num_position_iterations (default 16; 32 = accurate.yaml)")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher: start the N ranks as child processes (one per GPU,
    torch.distributed.run) BEFORE this process touches the GPU, and exit with the launcher's code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def cpu_baseline(sim_cfg_factory, n, seconds_budget=24.0):
    """The CPU oracle (scalar restatement of the same algorithm; 'port', not PhysX) on the host cores: same config,
    same action law, N = the benchmark's num_envs, on 4 threads (the reference's physx.num_threads,
    cfg/physics/default.yaml:20) and on all the cores this job may use (a one-GPU box hands the job a 16-core share of a
    256-thread host: more OpenMP threads than that only oversubscribe the share); bounded sample."""
    import numpy as np
    from oracle.oracle import Oracle
    sc, model = sim_cfg_factory(n)
    avail = len(os.sched_getaffinity(0))
    share = min(avail, int(os.environ.get("DEXSIM_BENCH_CPU_SHARE", "16")))
    runs = {}
    for cores in sorted({min(4, avail), share}):
        o = Oracle(sc, model.to_struct(), threads=cores)
        o.reset()
        rng = np.random.default_rng(1234)
        acts = [(2 * rng.random((n, 18)) - 1).astype(np.float32) for _ in range(4)]
        o.step(acts[0])
        t0, k = time.time(), 0
        while time.time() - t0 < seconds_budget / 2 and k < 200:
            o.step(acts[k % 4])
            k += 1
        runs[cores] = {"value": n * k / (time.time() - t0), "control_steps": k}
    best = max(runs, key=lambda c: runs[c]["value"])       # the fastest run is the baseline; the 4-thread figure stays next to it
    return {"value": runs[best]["value"], "unit": "env-steps/s", "cores": best, "kind": "port",
            "sample": f"BlindGrasping N={n}, {runs[best]['control_steps']} control steps, random actions, "
                      "oracle/dexsim_oracle.c with OpenMP (CPU restatement -- not PhysX; baseline, not target)",
            "threads_4": {"value": runs[min(runs)]["value"], "cores": min(runs), "control_steps": runs[min(runs)]["control_steps"]},
            "host_cores_visible": avail, "cpu_share_used": share}


def training_like(args, N, device, gen, steps=200, warm=60, keep_busy=None):
    """Secondary figure, NOT `value`: the regime a grasping policy trains in -- hands ON their boxes (hand/box and hand/ground
    contacts in every workgroup: the general contact path) AND de-synchronised episode clocks (some env resets in almost every
    control step: the device-gated second physics step runs) -- driven through the product surface make_env(...).step().
    Every reset re-creates the grasp-like pose: task.hand_translation_range is widened to 0.40 m and the reset samples are
    injected (DexSimCore.set_reset_samples, the parity tests' hook) so that a reset env lands with its hand base 0.247-0.262 m below
    the spawn pose, fingers at 0-0.3 rad: the fingertips of the fingers above the box rest / press on its top face (the
    state of tests/test_oracle_physics.py::test_fingers_press_on_box), the others hang beside it.  Actions: 0.2 x random (targets
    jitter about the pose).  Reports env-steps/s wall-clock, the contacts met and the resets per step."""
    import torch
    from dexrobot_isaac_amd import _abi
    from dexrobot_isaac_amd.config import default_cfg
    from dexrobot_isaac_amd.factory import make_env
    cfg = default_cfg("BlindGrasping")
    cfg["task"]["hand_translation_range"] = 0.40
    cfg["env"]["obsDict"] = args.obs_dict
    if args.joint_limit_rows:
        cfg["sim"]["dexsim_joint_limit_rows"] = True
    dev = str(device)
    env = make_env("BlindGrasping", N, dev, dev, 0, cfg=cfg)
    core = env._core
    u = torch.rand(N, _abi.NRESET_SAMPLES, device=device, generator=gen)
    u[:, 3:5] = 0.5 + 0.05 * (u[:, 3:5] - 0.5)                 # x, y offsets within +-1 cm of the box centre line
    u[:, 5] = 0.1725 + 0.01875 * u[:, 5]                         # z offset in [-0.262, -0.247] m: fingertips pressing on the box top (z = 0.05), as in the oracle's press test
    u[:, 6:9] = 0.5 + 0.1 * (u[:, 6:9] - 0.5)                  # hand rotation within +-0.04 rad
    u[:, 9] = 0.19 * u[:, 9]                                     # thumb rotation 0 .. 0.3 rad
    u[:, 10:] = 0.57 * u[:, 10:]                                 # other finger joints 0 .. 0.3 rad
    core.set_reset_samples(u)
    env.reset()
    es, tis = core.field("episode_step"), core.field("time_in_stage")
    k = torch.randint(0, 199, es.shape, device=device, generator=gen)
    es.copy_(k.to(es.dtype))
    tis.copy_(k.to(tis.dtype) * float(env.physics_manager.control_dt))
    acts = 0.2 * (2.0 * torch.rand(16, N, 18, device=device, generator=gen) - 1.0)
    for i in range(warm):
        env.step(acts[i % 16])
    if keep_busy is not None:
        keep_busy()                                               # (clock pre-roll, see main)
    r0 = float(core.field("reset_count").sum().item())
    for i in range(5):                                           # (a host read costs the first launches behind it ~80 us: five
        env.step(acts[i % 16])                                   # un-timed steps between the read and the window, see main)
    kb, kh = [], []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        env.step(acts[i % 16])
        if i % 20 == 19:                                         # (device scalars, no host sync)
            kb.append(core.stats[_abi.STAT["MEAN_CONTACTS"]].clone())
            kh.append(core.stats[_abi.STAT["MEAN_HAND_CONTACTS"]].clone())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = {"value": N * steps / dt, "unit": "env-steps/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "through": "make_env().step()",
           "resets_per_step": (float(core.field("reset_count").sum().item()) - r0) / (steps + 5),
           "mean_contacts": {"total": float(torch.stack(kb).mean().item()), "hand": float(torch.stack(kh).mean().item())},
           "state": "every reset lands hand-on-box (hand base 0.247-0.262 m below spawn: fingertips on the box top, fingers 0-0.3 rad; injected reset samples, "
                    "task.hand_translation_range = 0.40), episode clocks staggered over stage 1, actions 0.2 x U(-1,1)"}
    env.close()
    return res


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        self_launch(args)
    world = int(world_env or "1")
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
              f"(python bench.py --gpus N starts them itself)", file=sys.stderr)
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("DEXSIM_BENCH_LAUNCH_PROBE"):     # tests/test_bench_launch.py: the launch logic without a GPU
        # one write() per rank: two ranks printing through buffered streams can interleave inside a line
        os.write(1, f"probe rank {rank} local_rank {local_rank} world {world} master {os.environ.get('MASTER_ADDR')}\n".encode())
        sys.exit(0)

    import torch
    import torch.distributed as dist
    from dexrobot_isaac_amd import _abi
    from dexrobot_isaac_amd.config import build_sim_config, default_cfg
    from dexrobot_isaac_amd.core import DexSimCore
    from dexrobot_isaac_amd.rollout import RolloutBuffer

    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        if dist.get_world_size() != args.gpus:
            print(f"bench.py: RCCL world size {dist.get_world_size()} != --gpus {args.gpus}", file=sys.stderr)
            sys.exit(2)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    def factory(n):
        cfg = default_cfg(args.task)
        cfg["env"]["numEnvs"] = n
        if args.control_mode:
            cfg["task"]["controlMode"] = args.control_mode
        if args.joint_limit_rows:
            cfg["sim"]["dexsim_joint_limit_rows"] = True
        if args.substeps:
            cfg["sim"]["substeps"] = args.substeps
        if args.iterations:
            cfg["sim"]["physx"]["num_position_iterations"] = args.iterations
        dr = {"mass": (0.05, 0.2), "friction": (0.5, 1.5), "seed": 4242 + rank} if args.dr else None
        return build_sim_config(cfg, dr=dr)

    N = args.num_envs
    sc, model = factory(N)
    sc.seed = 42 + rank                      # rank-local reset stream
    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + rank)
    n_act = 64                               # distinct pre-generated action batches, cycled
    actions = 2.0 * torch.rand(n_act, N, 18, device=device, generator=gen) - 1.0
    # Optional pre-roll (--preroll P, stated in the output as config.preroll_steps): P control steps on a
    # THROW-AWAY env of the same shape immediately before each timed region's synchronisation.  Measured in round 3: no effect
    # (55.5 / 57.4 M with, 57.5 / 57.2 M without).  What a 20-step window behind 5 warm-up steps really paid for was a host read
    # (`.sum().item()`) between warm-up and window -- ~80 us on the first launches behind it (scripts/warm_profile.py: 61.1 us per
    # step GPU time without the read, 64.8 with); the read now sits before the warm-up.  Off by default.
    pre = None
    if args.preroll > 0:
        pre = DexSimCore(sc, model.to_struct(), device)
        pre.set_obs_dict_mode(args.obs_dict == "policy")
        pre.reset()

    def keep_busy():
        if pre is not None:
            for i in range(args.preroll):
                pre.step(actions[i % n_act])

    core = DexSimCore(sc, model.to_struct(), device)
    core.set_obs_dict_mode(args.obs_dict == "policy")
    core.reset()
    rollout = RolloutBuffer(args.horizon, N, sc.num_obs, device, mode=args.gather) if (world > 1 or args.rollout) else None

    pending, last_handle = [], [None]

    def run(k):
        for i in range(k):
            if rollout is not None:
                rollout.sink(core)           # this step's obs / rew / done (+ statistics block) land in the rollout slot: no copy kernels
            core.step(actions[i % n_act])
            if rollout is not None:
                if rollout.full():
                    # RCCL gather over xGMI (+ the all-reduce of the statistics rows), once per rollout, overlapped with the next
                    # rollout's simulation
                    pending.append(rollout.gather_async())
                    if len(pending) > 1:
                        last_handle[0] = pending.pop(0)
                        last_handle[0]()     # the previous rollout's gather must have landed by now
        while pending:
            last_handle[0] = pending.pop(0)
            last_handle[0]()

    # (read BEFORE the warm-up: a host read -- torch reduction kernel, device-to-host copy, wait -- between the warm-up and the
    # timed region costs the first launches behind it ~80 us, 4 us per step of a 20-step window: scripts/warm_profile.py, case B
    # against case A.  The resets counted therefore include the warm-up's.)
    resets0 = float(core.field("reset_count").sum().item())
    run(args.warmup)
    keep_busy()
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
    kbar = float(core.stats[_abi.STAT["MEAN_CONTACTS"]].item())
    kbar_hand = float(core.stats[_abi.STAT["MEAN_HAND_CONTACTS"]].item())

    collective = None
    if world > 1:
        gstats = last_handle[0].stats() if last_handle[0] is not None else None     # whole-population statistics of the last rollout (all ranks)
        # the end-of-rollout collective on its own (not overlapped), next to the simulation time of one rollout
        tms = []
        for _ in range(5):
            rollout.t = rollout.T
            torch.cuda.synchronize()
            dist.barrier()
            c0 = time.perf_counter()
            rollout.gather()
            torch.cuda.synchronize()
            tms.append((time.perf_counter() - c0) * 1e3)
        tt = torch.tensor([sorted(tms)[len(tms) // 2]], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        per_rank_mb = args.horizon * N * (sc.num_obs * 4 + 4 + 1) / 1e6
        collective = {"mode": args.gather, "collective_ms": float(tt.item()),
                      "global_stats_last_rollout": None if gstats is None else {
                          k: float(gstats[k].mean()) for k in ("success_rate", "failure_rate", "timeout_rate", "num_resets", "mean_contacts")},
                      "simulation_ms_per_rollout": dt / args.steps * 1e3 * args.horizon,
                      "payload_mb_per_rank": per_rank_mb, "rccl_ranks": dist.get_world_size(),
                      "note": "collective_ms = blocking gather alone (median of 5, max over ranks); in the timed region it is "
                              "issued asynchronously and overlaps the next rollout (double-buffered slots)"}

    if rank == 0:
        total_envs = N * world
        value = total_envs * args.steps / dt
        # The production launch (k_physics4<false>: actions + 4 sub-steps + post-physics) is issued once per control step,
        # so its mean duration over the timed region = GPU time between the two HIP events / steps: the figure also
        # contains the device-gated no-op launch (~3 us) and the dispatch gaps -- an upper bound.  A hipEvent pair around
        # each single launch is NOT used: the events' release/acquire fences make the kernel start from a cold L2.
        t_step = ev0.elapsed_time(ev1) * 1e3 / args.steps
        t_sub = core.time_stage(_abi.STAGE["SUBSTEP"], 50)     # a single sub-step (dynamics + solve + integrate + publish) as its own launch
        if args.joint_limit_rows:      # (the un-fused test kernels do not build joint-limit rows)
            t_solve = t_dyn = float("nan")
        else:
            t_solve = core.time_stage(_abi.STAGE["SOLVE"], 50)     # the same sweeps as a stand-alone kernel
            t_dyn = core.time_stage(_abi.STAGE["DYNAMICS"], 50)
        t_post = core.time_stage(_abi.STAGE["POST"], 20)
        t_pub = core.time_stage(_abi.STAGE["PUBLISH"], 20)
        b_sub = (632.0 + 204.0) * N                    # q, qd, targets, box in; q, qd, box out; cforce (last sub-step)
        b_dyn = (500.0 + 36.0 * kbar) * N
        pmc = {}
        try:                                           # HBM traffic from the separate rocprofv3 --pmc passes
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
        except Exception:
            pass

        def roof(kernel, bytes_, us, contacts=kbar, limited_by="dependent-issue latency (per-env sequential fp32 math)"):
            a = bytes_ / (us * 1e-6) / 1e9
            tr = None
            rec = pmc.get(kernel)
            if rec and int(rec.get("num_envs", -1)) == N:
                # gfx950: FETCH_SIZE tallies 64 B per 128-B request -> x2 (MI355X_MICROARCH.md, HBM); counters are in KB
                tr = (2.0 * rec["fetch_kb"] + rec["write_kb"]) * 1024.0
            # "bound" names the nominal roofline the fraction is taken against (HBM); at these fractions the
            # kernels are not bandwidth-bound -- `limited_by` says what actually limits them
            return {"bound": "hbm", "achieved": a, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": a / PEAK_HBM_GBS, "traffic": tr,
                    "kernel": kernel, "avg_us": us, "algorithmic_bytes": bytes_, "mean_contacts": contacts, "limited_by": limited_by}

        out = {
            "metric": f"env-steps/sec {args.task} num_envs={N} per MI355X (random actions)",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"{args.task} num_envs={N}/GPU" + (" (BASELINE configs[2]; configs[3] when n_gpus>1)"
                                    if args.task == "BlindGrasping" and N == 4096 and not args.dr else "")),
                       "control_mode": args.control_mode or "task default",
                       "num_envs_per_gpu": N, "sim_dt": float(sc.dt), "substeps": int(sc.substeps), "pgs_iterations": int(sc.num_position_iterations),
                       "preroll_steps": args.preroll,
                       "obs_dict": ("policy keys only (env.obsDict = policy: the 392 SoA rows behind get_observations_dict() are not materialised)"
                                    if args.obs_dict == "policy" else "all 392 rows materialised every step (env.obsDict = all, the product default)"),
                       "contact_solver": "warm-started block-parallel PGS with mass splitting (HIP: csrc/dexsim_physics.hip.inc, phases 3-4 of substep_body)",
                       "parallelism": f"env-shard x{world}", "rollout_gather_horizon": args.horizon if world > 1 else None,
                       "rollout_gather_mode": args.gather if world > 1 else None,
                       "domain_randomisation": bool(args.dr), "joint_limit_rows": bool(args.joint_limit_rows)},
            "resets_per_step": resets / (args.steps + args.warmup),      # (over warm-up + timed steps, see resets0)
            # the headline regime's contact set: random actions from the spawn pose never bring a finger to the box, so
            # these are the box's corners on the ground (see contact_rich for the regime the solver exists for)
            "mean_contacts": {"total": kbar, "hand": kbar_hand, "box_ground": kbar - kbar_hand},
            # dominant kernel = the one launch that carries a whole control step (k_physics4<false> with the action and
            # post-physics blocks); algorithmic bytes = SURVEY 8d's whole-env-step figure
            "roofline": roof("k_physics4", STEP_BYTES * N, t_step),
            "roofline_substep": roof("k_substep", b_sub, t_sub),
            # (the stand-alone k_solve in the headline state: 4 box/ground contacts, no hand contact -- kept for continuity with
            # rounds 1-2; the BASELINE sub-metric on the PRODUCTION kernel is roofline_contact_solve below, filled by the
            # contact_rich leg from the kernel's own phase stamps)
            "roofline_contact_solve_isolated_kernel": roof("k_solve", (256.0 + 60.0 * kbar) * N, t_solve),
            "roofline_dynamics": roof("k_dynamics", b_dyn, t_dyn),
            "roofline_whole_step": {"bound": "hbm", "achieved": STEP_BYTES * value / world / 1e9, "peak": PEAK_HBM_GBS,
                                    "unit": "GB/s", "frac": STEP_BYTES * value / world / 1e9 / PEAK_HBM_GBS, "traffic": None},
            "kernel_us": {"k_physics4_step": t_step, "k_substep": t_sub, "k_dynamics": t_dyn, "k_solve": t_solve, "k_post": t_post, "k_publish": t_pub},
        }
        if collective is not None:
            out["rollout_gather"] = collective
        if world == 1 and not args.no_stagger:
            # Secondary figure, NOT `value`: the timed region above starts from a fresh env, like the reference's own run,
            # so all episodes end in the same control step and the conditional second physics step of
            # reset_manager.py:180 runs once per ~200 steps.  In training the episodes de-synchronise and some env
            # resets in (almost) every control step; this run staggers the episode clocks to show that regime.
            # (random actions end every episode through the stage-1 pre-grasp check at t = 4 s = 200 control steps, so the
            # stage clock and the episode clock are staggered together)
            es, tis = core.field("episode_step"), core.field("time_in_stage")
            k = torch.randint(0, 199, es.shape, device=device, generator=gen)
            es.copy_(k.to(es.dtype))
            tis.copy_(k.to(tis.dtype) * float(sc.control_dt))
            rollout = None                   # (the secondary figures are measured without the rollout sink)
            core.set_step_sink(None, None, None)
            run(50)
            keep_busy()
            r0 = float(core.field("reset_count").sum().item())
            run(5)                           # (un-timed: the launches right behind a host read run slow, see resets0)
            torch.cuda.synchronize()
            ts = time.perf_counter()
            run(200)
            torch.cuda.synchronize()
            ds = time.perf_counter() - ts
            out["staggered_resets"] = {"value": N * 200 / ds, "unit": "env-steps/s", "ms_per_step": ds / 200 * 1e3,
                                       "resets_per_step": (float(core.field("reset_count").sum().item()) - r0) / 205}
        if world == 1 and not args.no_contact_rich and args.task == "BlindGrasping":
            # Secondary figure, NOT `value`: every hand lowered onto its box (scripts/contact_regime.py's state) -- the
            # regime a policy that has learnt to grasp lives in.  Timed: back-to-back physics steps (4 sub-steps each,
            # general contact path in every workgroup), HIP events around the region; and the stand-alone contact-solve
            # kernel in that state for the BASELINE sub-metric "contact-solve HBM %".
            q = core.field("q")
            q.zero_()
            q[2] = -0.40
            q[6:] = 0.3 * torch.rand(20, N, device=device, generator=gen)
            core.field("qd").zero_()
            core.field("targets").copy_(q)
            for _ in range(20):
                core.physics_step(False)
            torch.cuda.synchronize()
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
            for _ in range(100):      # (the same 20 + 100 physics steps as scripts/contact_regime.py)
                core.physics_step(False)
            c1.record()
            torch.cuda.synchronize()
            us_phys = c0.elapsed_time(c1) * 1e3 / 100
            nc = core.field("ncontact").float()
            k_mean, k_max = float(nc.mean().item()), int(nc.max().item())
            # Those 100 steps follow the teleport by only 20 (round 1's protocol, kept so that 769 -> ... stays comparable): the
            # contact lists are still settling and part of the workgroups run the generic variant of the sweeps.  The settled
            # state -- the regime itself -- is the next 100 steps.
            probe = core.set_phase_probe(True)     # shader-clock stamps of the production kernel's solver phases (dexsim_set_phase_probe)
            c0.record()
            for _ in range(100):
                core.physics_step(False)
            c1.record()
            torch.cuda.synchronize()
            us_settled = c0.elapsed_time(c1) * 1e3 / 100
            import numpy as np
            pr = probe.cpu().numpy().view(np.uint32).astype(np.float64)[: (N + 63) // 64]
            core.set_phase_probe(False)
            ncs = core.field("ncontact").float()
            code = core.field("ccode")
            kidx = torch.arange(code.shape[0], device=device)[:, None]
            hand = float((((code & 3) < 2) & (kidx < core.field("ncontact"))).sum().item()) / N
            # BASELINE sub-metric "contact-solve HBM %" on the PRODUCTION kernel: phases 3 + 4 of the general contact path (contact rows
            # + the block solver's 17 passes) as a fraction of the launch, from the kernel's own s_memtime stamps, times the launch
            # time between HIP events; settled contact-rich state, hand contacts > 0.  Next to it the bound that actually binds:
            # VALU issue (SQ_INSTS_VALU x 4 cycles per wave64 instruction over SIMD-cycles), chip-wide and on the occupied CUs.
            k_set = float(ncs.mean().item())
            f34, f4 = float(pr[:, 0].sum() / pr[:, 1].sum()), float(pr[:, 3].sum() / pr[:, 1].sum())
            ticks_launch = float(np.median(pr[:, 1])) / 100.0
            solver_us, sweeps_us = us_settled * f34, us_settled * f4
            solve_bytes = (256.0 + 60.0 * k_set) * N * 4
            issue = (pmc.get("issue") or {}).get("contact_settled") if int((pmc.get("issue") or {}).get("num_envs", -1)) == N else None
            valu = None
            if issue:
                occ_simds = 4 * min((N + 63) // 64, 256)
                valu = {"SQ_INSTS_VALU_per_launch": issue["SQ_INSTS_VALU"], "kernel_cycles": ticks_launch,
                        "valu_issue_frac_chip": issue["SQ_INSTS_VALU"] * 4.0 / (1024 * ticks_launch),
                        "valu_issue_frac_occupied_cus": issue["SQ_INSTS_VALU"] * 4.0 / (occ_simds * ticks_launch),
                        "source": "profiles/pmc_latest.json (rocprofv3 --pmc SQ_INSTS_VALU, own pass) / cycles from the in-run probe"}
            out["roofline_contact_solve"] = {
                "bound": "hbm", "achieved": solve_bytes / (solver_us * 1e-6) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": solve_bytes / (solver_us * 1e-6) / 1e9 / PEAK_HBM_GBS, "traffic": None,
                "kernel": "k_physics4<false>, phases 3-4 of substep_body (general contact path), 4 sub-steps", "avg_us": solver_us,
                "algorithmic_bytes": solve_bytes, "mean_contacts": k_set,
                "solver_share_of_launch": f34, "sweeps_only_us": sweeps_us, "sweeps_share_of_launch": f4,
                "general_path_substeps_per_launch": float(pr[:, 2].sum() / pr.shape[0] / 100.0),
                "shader_clock_mhz_from_probe": ticks_launch / us_settled, "valu_issue": valu,
                "limited_by": "dependent fp32 issue on 64 of 256 CUs (N = 4096 is 64 workgroups): row build chains + 17 passes x (contact update + one barrier)"}
            ih = (pmc.get("issue") or {}).get("headline") if int((pmc.get("issue") or {}).get("num_envs", -1)) == N else None
            if ih and "SQ_INSTS_VALU" in ih:      # the same bound for the headline launch (cycles: launch time x the probe's clock)
                cyc = t_step * ticks_launch / us_settled
                occ_simds = 4 * min((N + 63) // 64, 256)
                out["roofline"]["valu_issue"] = {"valu_issue_frac_chip": ih["SQ_INSTS_VALU"] * 4.0 / (1024 * cyc),
                                                 "valu_issue_frac_occupied_cus": ih["SQ_INSTS_VALU"] * 4.0 / (occ_simds * cyc),
                                                 "SQ_INSTS_VALU_per_launch": ih["SQ_INSTS_VALU"], "kernel_cycles": cyc,
                                                 "source": "profiles/pmc_latest.json (own rocprofv3 --pmc pass) / cycles = avg_us x shader clock from the in-run probe"}
            out["contact_rich"] = {
                "state": "hand base lowered 0.40 m onto the box in every env, fingers at U(0,0.3) rad, targets = pose; 20 physics steps to develop the contacts",
                "us_per_physics_step": us_phys, "env_steps_per_s_physics_only": N / (us_phys * 1e-6),
                "timed": "steps 21-120 after the teleport (round 1's protocol; lists still settling)",
                "settled": {"us_per_physics_step": us_settled, "env_steps_per_s_physics_only": N / (us_settled * 1e-6),
                            "timed": "steps 121-220", "mean_contacts": float(ncs.mean().item()), "max_contacts": int(ncs.max().item())},
                "mean_contacts": {"total": k_mean, "hand": hand, "box_ground": k_mean - hand}, "max_contacts": k_max,
                # BASELINE sub-metric "contact-solve HBM %" in this regime, for the fused production path: 4 sub-steps per physics
                # step, solver bytes only.  (The stand-alone k_solve is a one-wave test kernel for the general path since round 2's
                # block solver; it is timed only in the headline state, where all contacts are box/ground.)
                "roofline_contact_solve_fused": {"bound": "hbm", "achieved": (256.0 + 60.0 * k_mean) * N * 4 / (us_phys * 1e-6) / 1e9,
                                                 "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                                 "frac": (256.0 + 60.0 * k_mean) * N * 4 / (us_phys * 1e-6) / 1e9 / PEAK_HBM_GBS,
                                                 "limited_by": "per-contact update chains of the block solver (17 passes x ~300 dependent instructions) and their exchange barriers",
                                                 "note": "upper bound on the solver's share: the whole physics step's time is charged to it"},
            }
        if world == 1 and not args.no_training_like and args.task == "BlindGrasping":
            out["training_like"] = training_like(args, N, device, gen, keep_busy=keep_busy)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(factory, N)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
