"""Child process of test_debug_spin_build_never_hits_its_bound: drives the library named by DEXSIM_LIB_PATH (the product build
or its -DDEXSIM_DEBUG_SPIN twin) through the headline regime (hand clear: Schur-split tokens, broadphase tokens) and the
contact-rich regime (general contact path) and prints one JSON line: the spin-timeout word of the counters block and a digest of
the outputs."""
import hashlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dexrobot_isaac_amd.config import build_sim_config, default_cfg  # noqa: E402
from dexrobot_isaac_amd.core import DexSimCore  # noqa: E402

CNT_SPIN_TIMEOUT = 20     # csrc/dexsim_device.h

n = 320
cfg = default_cfg("BlindGrasping")
cfg["env"]["numEnvs"] = n
sc, model = build_sim_config(cfg)
core = DexSimCore(sc, model.to_struct(), "cuda:0")
core.reset()
g = torch.Generator(device="cuda:0").manual_seed(7)
h = hashlib.sha256()
for _ in range(12):
    core.step(2 * torch.rand(n, 18, device="cuda:0", generator=g) - 1)
h.update(core.obs_buf.cpu().numpy().tobytes())
q = core.field("q")
q.zero_()
q[2] = -0.40
q[6:] = 0.3 * torch.rand(20, n, device="cuda:0", generator=g)
core.field("qd").zero_()
core.field("targets").copy_(q)
for _ in range(12):
    core.physics_step(False)
torch.cuda.synchronize()
h.update(core.field("q").cpu().numpy().tobytes())
h.update(core.field("box_pos").cpu().numpy().tobytes())
print(json.dumps({"spin_word": int(core.counters[CNT_SPIN_TIMEOUT].item()), "digest": h.hexdigest(),
                  "hand_contacts": float((core.field("ncontact").float().mean()).item())}))
