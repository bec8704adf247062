#!/usr/bin/env python
"""Dev probe: staggered-reset regime (bench.py's secondary figure) with a variant library (DEXSIM_LIB)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("DEXSIM_LIB"):
    from dexrobot_isaac_amd import _lib
    _lib.LIB_PATH = os.environ["DEXSIM_LIB"]
from dexrobot_isaac_amd.config import build_sim_config, default_cfg
from dexrobot_isaac_amd.core import DexSimCore
n = 4096
cfg = default_cfg("BlindGrasping"); cfg["env"]["numEnvs"] = n
sc, model = build_sim_config(cfg)
core = DexSimCore(sc, model.to_struct(), "cuda:0")
core.reset()
g = torch.Generator(device="cuda:0").manual_seed(1)
acts = [2 * torch.rand(n, 18, device="cuda:0", generator=g) - 1 for _ in range(64)]
def run(k):
    for i in range(k): core.step(acts[i % 64])
run(50); torch.cuda.synchronize()
t = time.perf_counter(); run(300); torch.cuda.synchronize(); print("headline us/step", (time.perf_counter() - t) / 300 * 1e6)
import statistics
def bursts(label):
    for burst in (1, 4, 16):
        ts = []
        for rep in range(30):
            torch.cuda.synchronize(); t = time.perf_counter(); run(burst); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e6)
        print(f"{label}: burst of {burst} steps after a sync: median {statistics.median(ts):.0f} us  ({statistics.median(ts) / burst:.0f} us/step)")
bursts("headline regime")
es, tis = core.field("episode_step"), core.field("time_in_stage")
k = torch.randint(0, 199, es.shape, device="cuda:0", generator=g)
es.copy_(k.to(es.dtype)); tis.copy_(k.to(tis.dtype) * float(sc.control_dt))
run(50); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t = time.perf_counter(); e0.record(); run(200); e1.record(); torch.cuda.synchronize()
nc = core.field("ncontact").float(); print("contacts mean", nc.mean().item(), "max", nc.max().item()); print("staggered us/step wall", (time.perf_counter() - t) / 200 * 1e6, "events", e0.elapsed_time(e1) * 1e3 / 200)

for chunk in (5, 20, 50):
    torch.cuda.synchronize(); t = time.perf_counter()
    for c in range(200 // chunk):
        run(chunk); torch.cuda.synchronize()
    print(f"staggered, synchronised every {chunk} steps: wall us/step", (time.perf_counter() - t) / 200 * 1e6)
import ctypes
t = time.perf_counter(); run(200); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("issue time us/step", (t1 - t) / 200 * 1e6, "then drain", (t2 - t1) * 1e6)

bursts("staggered regime")
