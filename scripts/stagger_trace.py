#!/usr/bin/env python
"""Dev probe: the staggered-reset regime (some env resets in every control step: both launches of a step are full physics
steps) for a kernel trace:  rocprofv3 --kernel-trace --output-format csv -d <dir> -o s -- python3 scripts/stagger_trace.py
then  python scripts/stagger_trace.py --analyse <dir>  prints per-launch durations and the gaps between consecutive launches."""
import glob
import os
import sys

if len(sys.argv) > 2 and sys.argv[1] == "--analyse":
    import pandas as pd
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    d = pd.read_csv(f).sort_values("Start_Timestamp")
    d = d[d["Kernel_Name"].str.contains("k_physics4", regex=False)].tail(200).reset_index(drop=True)
    d["dur"] = (d["End_Timestamp"] - d["Start_Timestamp"]) / 1e3
    d["gap"] = (d["Start_Timestamp"] - d["End_Timestamp"].shift(1)) / 1e3
    d["gated"] = d["Kernel_Name"].str.contains("<true>", regex=False) | d["Kernel_Name"].str.contains("(bool)1", regex=False)
    for g, name in ((False, "k_physics4<false>"), (True, "k_physics4<true> (gated)")):
        x = d[d["gated"] == g]
        print(f"{name}: n {len(x)}  duration mean {x['dur'].mean():.1f} us (min {x['dur'].min():.1f}, max {x['dur'].max():.1f})  gap before it mean {x['gap'].mean():.1f} us (max {x['gap'].max():.1f})")
    span = (d["End_Timestamp"].iloc[-1] - d["Start_Timestamp"].iloc[0]) / 1e3
    print(f"span of the last {len(d)} launches: {span:.0f} us = {span / (len(d) / 2):.1f} us per control step; kernels {d['dur'].sum() / (len(d) / 2):.1f}, gaps {d['gap'].iloc[1:].sum() / (len(d) / 2):.1f}")
    sys.exit(0)

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dexrobot_isaac_amd.config import build_sim_config, default_cfg  # noqa: E402
from dexrobot_isaac_amd.core import DexSimCore  # noqa: E402

n = 4096
cfg = default_cfg("BlindGrasping")
cfg["env"]["numEnvs"] = n
sc, model = build_sim_config(cfg)
core = DexSimCore(sc, model.to_struct(), "cuda:0")
core.set_obs_dict_mode(True)
core.reset()
g = torch.Generator(device="cuda:0").manual_seed(1)
acts = 2 * torch.rand(64, n, 18, device="cuda:0", generator=g) - 1
es, tis = core.field("episode_step"), core.field("time_in_stage")
k = torch.randint(0, 199, es.shape, device="cuda:0", generator=g)
es.copy_(k.to(es.dtype))
tis.copy_(k.to(tis.dtype) * float(sc.control_dt))
for i in range(60):
    core.step(acts[i % 64])
torch.cuda.synchronize()
for i in range(120):
    core.step(acts[i % 64])
torch.cuda.synchronize()
print("done")
