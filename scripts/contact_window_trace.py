#!/usr/bin/env python
"""The protocol window of the contact-rich leg (steps 21-120 after the teleport) step by step: GPU time per physics step and the
length of the contact lists (KMAX = 24 entries per env).  (The host reads every 10 steps cost a few us per step: the absolute times
read ~10 % above scripts/contact_regime.py's.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("DEXSIM_LIB"):
    from dexrobot_isaac_amd import _lib
    _lib.LIB_PATH = os.environ["DEXSIM_LIB"]
from dexrobot_isaac_amd.config import build_sim_config, default_cfg  # noqa: E402
from dexrobot_isaac_amd.core import DexSimCore  # noqa: E402

n = 4096
z = float(sys.argv[1]) if len(sys.argv) > 1 else -0.40
cfg = default_cfg("BlindGrasping")
cfg["env"]["numEnvs"] = n
sc, model = build_sim_config(cfg)
core = DexSimCore(sc, model.to_struct(), "cuda:0")
core.reset()
g = torch.Generator(device="cuda:0").manual_seed(3)
for zz in (0.0, z):       # (the same sequence of states as scripts/contact_regime.py up to this regime)
    q = core.field("q")
    q.zero_()
    q[2] = zz
    q[6:] = 0.3 * torch.rand(20, n, device="cuda:0", generator=g)
    core.field("qd").zero_()
    core.field("targets").copy_(q)
    if zz == 0.0:
        for _ in range(120):
            core.physics_step(False)
T = 220
ev = [torch.cuda.Event(enable_timing=True) for _ in range(T + 1)]
rec = []
ev[0].record()
for t in range(T):
    core.physics_step(False)
    ev[t + 1].record()
    if t % 10 == 9:
        nc = core.field("ncontact").to(torch.int64)
        rec.append((t + 1, nc.float().mean(), nc.max(), (nc.view(-1, 64).max(1).values >= 16).sum()))
torch.cuda.synchronize()
print(f"z offset {z}: per block of 10 physics steps (the stats are those of the block's last step)")
print("steps      us/step  contacts/env(mean max)  WGs with a lane >= 16 contacts (> 12 hand contacts: the streamed sweep variant)")
for i, r in enumerate(rec):
    us = ev[10 * i].elapsed_time(ev[10 * i + 10]) * 100
    print(f"{r[0] - 9:3d}-{r[0]:3d}  {us:8.1f}   {float(r[1]):5.2f} {int(r[2]):3d}          {int(r[3]):3d}")
