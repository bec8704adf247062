#!/usr/bin/env python
"""Why a 20-step window behind 5 warm-up steps reads slower than a 200-step one: the GPU time (HIP events) of short windows of
control steps behind different preambles -- warm-up length, a host read in between, idle time, windows back to back."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dexrobot_isaac_amd.config import build_sim_config, default_cfg  # noqa: E402
from dexrobot_isaac_amd.core import DexSimCore  # noqa: E402

n = 4096
cfg = default_cfg("BlindGrasping")
cfg["env"]["numEnvs"] = n
sc, model = build_sim_config(cfg)
core = DexSimCore(sc, model.to_struct(), "cuda:0")
core.set_obs_dict_mode(1)
g = torch.Generator(device="cuda:0").manual_seed(1234)
acts = 2.0 * torch.rand(64, n, 18, device="cuda:0", generator=g) - 1.0
step_i = [0]


def steps(k):
    for _ in range(k):
        core.step(acts[step_i[0] % 64])
        step_i[0] += 1


def window(k=20, per=5):
    nb = k // per
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(nb + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for b in range(nb):
        steps(per)
        ev[b + 1].record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e6 / k
    blocks = [ev[b].elapsed_time(ev[b + 1]) * 1e3 / per for b in range(nb)]
    return f"wall {wall:.1f} gpu {ev[0].elapsed_time(ev[nb]) * 1e3 / k:.1f} us/step; blocks of {per}: " + " ".join(f"{x:.1f}" for x in blocks)


def fresh():
    core.reset()
    step_i[0] = 0
    torch.cuda.synchronize()


fresh(); time.sleep(1.0)
steps(5); torch.cuda.synchronize()
print("A  1 s idle, warm 5, sync:               ", window())
fresh(); time.sleep(1.0)
steps(5); x = float(core.field("reset_count").sum().item()); torch.cuda.synchronize()
print("B  1 s idle, warm 5, .item(), sync:      ", window())
fresh(); time.sleep(1.0)
steps(50); torch.cuda.synchronize()
print("C  1 s idle, warm 50, sync:              ", window())
fresh(); time.sleep(1.0)
steps(5); torch.cuda.synchronize(); time.sleep(0.01)
print("D  1 s idle, warm 5, sync, 10 ms idle:   ", window())
fresh(); time.sleep(1.0)
print("E  1 s idle, no warm-up:                 ", window())
fresh()
steps(5); torch.cuda.synchronize()
print("F  no idle, warm 5, sync:                ", window())
for i in range(3):
    print(f"G{i} next window directly behind the sync:  ", window())
fresh(); time.sleep(1.0)
steps(5); torch.cuda.synchronize()
print("H  1 s idle, warm 5, sync, 200 steps:    ", window(200, 20))
