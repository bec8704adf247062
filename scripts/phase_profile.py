#!/usr/bin/env python
"""Diagnostic: builds libdexsim with -DDEXSIM_PROFILE_PHASES into a scratch path and prints where a sub-step's cycles go
(per wave, s_memtime at the phase boundaries).  The product library is never built with this flag."""
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dexrobot_isaac_amd import _abi, _lib  # noqa: E402
from dexrobot_isaac_amd.build import CSRC  # noqa: E402

out = os.path.join(ROOT, "gpurun_out", "libdexsim_prof.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-fno-hip-fp32-correctly-rounded-divide-sqrt", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-fPIC", "-shared", "-std=c++17", "-DDEXSIM_PROFILE_PHASES"] +
                      os.environ.get("DEXSIM_EXTRA_DEFS", "").split() + [
                       "-o", out, os.path.join(CSRC, "dexsim.hip")], cwd=CSRC)   # DEXSIM_EXTRA_DEFS: experiment switches (-DEXP_...)
_lib.LIB_PATH = out
from dexrobot_isaac_amd.config import build_sim_config, default_cfg  # noqa: E402
from dexrobot_isaac_amd.core import DexSimCore  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = default_cfg("BlindGrasping")
cfg["env"]["numEnvs"] = n
sc, model = build_sim_config(cfg)
core = DexSimCore(sc, model.to_struct(), "cuda:0")
core.reset()
a = 2 * torch.rand(n, 18, device="cuda:0") - 1
if len(sys.argv) <= 2:
    for _ in range(10):
        core.step(a)
if len(sys.argv) > 2:   # contact-rich regime: python scripts/phase_profile.py 4096 -0.40  (scripts/contact_regime.py's state)
    g = torch.Generator(device="cuda:0").manual_seed(3)
    for z, nsteps in ((0.0, 120), (float(sys.argv[2]), 120)):     # the same sequence of states as scripts/contact_regime.py
        q = core.field("q")
        q.zero_()
        q[2] = z
        q[6:] = 0.3 * torch.rand(20, n, device="cuda:0", generator=g)
        core.field("qd").zero_()
        core.field("targets").copy_(q)
        for _ in range(nsteps):
            core.physics_step(False)
    print("contact regime: contacts/env mean %.2f max %d" % (core.field("ncontact").float().mean().item(), int(core.field("ncontact").max().item())))
core.run_stage(_abi.STAGE["SUBSTEP"])
torch.cuda.synchronize()
crow = core.field("crow").view(torch.int32).cpu().numpy()      # stamps of lane 0 of every block: rows wv*8 + k
names = ["base chain", "phase1 fingers|palm", "phase2 schur|narrow", "phase3 rows", "phase4 sweeps", "phase5 integrate", "publish",
         "(general path) end of sweep 1 / wave 6: of its first box block"]
lanes = np.arange(0, n, 64)
print(f"N={n}: cycles since kernel start at each boundary, median over {len(lanes)} workgroups (s_memtime @100 MHz ticks x?)")
for wv in range(7):
    st = np.array([[crow[wv * 8 + k, e] for k in range(8)] for e in lanes])
    med = np.median(st, axis=0)
    print(f"wave {wv}: " + "  ".join(f"{nm}={int(v)}" for nm, v in zip(names, med)))
    if wv == 0 and "DBG_SCHUR" in os.environ.get("DEXSIM_EXTRA_DEFS", ""):   # inside wave 0's Schur phase
        ss = np.median(np.array([[crow[56 + k, e] for k in range(6)] for e in lanes]), axis=0)
        print("  wave 0, Schur phase: start %d, composite sum %d, own rows %d, tokens seen %d, S / tau_B complete %d, solved %d" % tuple(int(v) for v in ss))
    if wv == 0 and "DBG_PH2" in os.environ.get("DEXSIM_EXTRA_DEFS", ""):   # inside phase 2 of the general path: waves 1 and 5
        for w, off in ((1, 56), (5, 60)):
            ss = np.median(np.array([[crow[off + k, e] for k in range(3)] for e in lanes]), axis=0)
            print(f"  wave {w}, phase 2: narrowphase done at {int(ss[0])}, barrier passed {int(ss[1])}, appended {int(ss[2])}")
    elif wv == 0:   # inside pass 2 of the general path's sweeps: waves 0 and 5
        for w, off in ((0, 56), (5, 61)):
            ss = np.median(np.array([[crow[off + k, e] for k in range(5)] for e in lanes]), axis=0)
            print(f"  wave {w}, sweep 2: starts at {int(ss[0])}, item done +{int(ss[1]-ss[0])}, barrier passed +{int(ss[2]-ss[1])}, reduction done +{int(ss[3]-ss[2])}, second barrier passed +{int(ss[4]-ss[3])}")
    if wv == 0:   # the launch ends with its slowest workgroup
        slow = st[np.argmax(st[:, 6])]
        print("wave 0, slowest workgroup: " + "  ".join(f"{nm}={int(v)}" for nm, v in zip(names, slow)))
        print("wave 0, publish stamp over workgroups: min %d  median %d  p90 %d  max %d" % (st[:, 6].min(), np.median(st[:, 6]), np.percentile(st[:, 6], 90), st[:, 6].max()))

if "DBG_SWEEP" in os.environ.get("DEXSIM_EXTRA_DEFS", ""):   # pass 2 of the sweeps, every wave (rows 400 + 8 wave + i)
    print("pass 2 of the sweeps, per wave (median workgroup): start | own work done | first barrier passed | reduction done | second barrier passed")
    for wv in range(7):
        ss = np.median(np.array([[crow[400 + wv * 8 + k, e] for k in range(5)] for e in lanes]), axis=0)
        print(f"  wave {wv}: start {int(ss[0])}  work +{int(ss[1]-ss[0])}  B1 at {int(ss[2])} (+{int(ss[2]-ss[1])})  reduce +{int(ss[3]-ss[2])}  B2 at {int(ss[4])} (+{int(ss[4]-ss[3])})")

# k_post: per-wave stamps after phase A (helper tasks / wave-0 loads), phase B (wave-0 logic), the row flush and the
# obs_buf flush; then wave 0's logic split (relative to the start of phase B)
core.run_stage(_abi.STAGE["POST"])
torch.cuda.synchronize()
crow = core.field("crow").view(torch.int32).cpu().numpy()
pn = ["phase A", "phase B", "row flush", "obs_buf flush"]
for wv in range(8):
    st = np.array([[crow[wv * 8 + k, e] for k in range(4)] for e in lanes])
    print(f"k_post wave {wv}: " + "  ".join(f"{nm}={int(v)}" for nm, v in zip(pn, np.median(st, axis=0))))
st = np.array([[crow[32 + k, e] for k in range(2, 7)] for e in lanes])
print("k_post logic (wave 0, from phase B start): " + "  ".join(f"{nm}={int(v)}" for nm, v in zip(
    ["task obs+FSM", "termination", "reward terms", "reward table", "stats"], np.median(st, axis=0))))

# the production launch: four bodies back to back (stamps are relative to each body's own start)
core.run_stage(_abi.STAGE["PHYSICS"])
torch.cuda.synchronize()
crow = core.field("crow").view(torch.int32).cpu().numpy()
for b in range(4):
    for wv in (0, 4, 5, 6):
        st = np.array([[crow[64 * (b + 1) + wv * 8 + k, e] for k in range(7 if b == 3 else 6)] for e in lanes])
        print(f"k_physics4 body {b} wave {wv}: " + "  ".join(f"{nm}={int(v)}" for nm, v in zip(names, np.median(st, axis=0))))

# what a stamp tick is worth: the same launch timed with events (isolated, then 20 back to back), next to the sum of the four
# bodies' stamps of the median workgroup
tot = 0
for b in range(4):
    st = np.array([[crow[64 * (b + 1) + 0 * 8 + k, e] for k in range(7 if b == 3 else 6)] for e in lanes])
    tot += int(np.median(st, axis=0)[-1])
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); core.run_stage(_abi.STAGE["PHYSICS"]); e1.record(); torch.cuda.synchronize()
one = e0.elapsed_time(e1) * 1e3
e0.record()
for _ in range(20):
    core.run_stage(_abi.STAGE["PHYSICS"])
e1.record(); torch.cuda.synchronize()
many = e0.elapsed_time(e1) * 1e3 / 20
print(f"k_physics4 (profile build): {tot} stamp ticks over the four bodies (wave 0, median workgroup); one launch {one:.1f} us, 20 back to back {many:.1f} us each -> {tot / many / 1e3:.2f} ticks per ns")

# the whole control step as the product runs it (actions + 4 bodies + post block in one launch): the last body's stamps for every
# wave -- who reaches the barrier in front of the post block last -- and the post block's own (relative to its start)
core.step(a)
torch.cuda.synchronize()
crow = core.field("crow").view(torch.int32).cpu().numpy()
print("full step, last body (median workgroup): phase 5 done | publication (+ the box wave's post pre-tasks) done")
for wv in range(7):
    st = np.median(np.array([[crow[64 * 4 + wv * 8 + k, e] for k in range(7)] for e in lanes]), axis=0)
    print(f"  wave {wv}: {int(st[5])} | {int(st[6])}")
pn = ["phase A", "phase B", "row flush", "obs_buf flush"]
for wv in range(7):
    st = np.array([[crow[wv * 8 + k, e] for k in range(4)] for e in lanes])
    print(f"  post block wave {wv}: " + "  ".join(f"{nm}={int(v)}" for nm, v in zip(pn, np.median(st, axis=0))))
