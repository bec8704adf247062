#!/usr/bin/env python
"""Per-kernel timing (HIP events, dexsim_time_stage) of one BlindGrasping instance; dev tool for the GPU box."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dexrobot_isaac_amd import _abi  # noqa: E402
from dexrobot_isaac_amd.config import build_sim_config, default_cfg  # noqa: E402
from dexrobot_isaac_amd.core import DexSimCore  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--num-envs", type=int, nargs="+", default=[4096])
ap.add_argument("--launches", type=int, default=50)
ap.add_argument("--iters", type=int, nargs="+", default=[16])
ap.add_argument("--task", default="BlindGrasping")
args = ap.parse_args()
import itertools
for n, iters in itertools.product(args.num_envs, args.iters):
    cfg = default_cfg(args.task)
    cfg["env"]["numEnvs"] = n
    cfg["sim"]["physx"]["num_position_iterations"] = iters
    sc, model = build_sim_config(cfg)
    core = DexSimCore(sc, model.to_struct(), "cuda:0")
    core.reset()
    a = 2 * torch.rand(n, 18, device="cuda:0") - 1
    for _ in range(20):
        core.step(a)
    torch.cuda.synchronize()
    out = {k: round(core.time_stage(v, args.launches), 2) for k, v in _abi.STAGE.items() if k not in ("FINALIZE",)}
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        core.step(a)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 100
    print(f"N={n} iters={iters}: us/launch {out}  step {dt * 1e6:.1f} us  -> {n / dt / 1e6:.2f} M env-steps/s  contacts {float(core.stats[18]):.2f}")
    core.close()
