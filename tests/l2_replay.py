"""Replay of the golden L2 scenarios (tests/golden/l2_*.npz, produced by the reference's own Python) against
a backend exposing the oracle-style stage interface.  Used with the CPU oracle (-m "not gpu") and with the
HIP kernels through the C-ABI (-m gpu)."""
import json

import numpy as np

from dexrobot_isaac_amd.config import build_sim_config, default_cfg

L1_FIELDS = ("q", "qd", "site_pose", "hand_vel", "cforce", "box_pos", "box_quat", "box_lin", "box_ang")


def scenario_config(npz, num_envs=None):
    cfg = default_cfg(str(npz["task"]))
    for k, v in json.loads(str(npz["cfg_overrides"])).items():
        d = cfg
        parts = k.split(".")
        for p in parts[:-1]:
            d = d[p]
        d[parts[-1]] = v
    cfg["env"]["numEnvs"] = int(npz["N"]) if num_envs is None else num_envs
    return cfg


def inject(backend, npz, prefix, t=None):
    for f in L1_FIELDS:
        key = f"{prefix}_{f}"
        if key in npz.files:
            arr = npz[key] if t is None else npz[key][t]
            backend.set(f, arr)


def replay(backend, npz, atol=2e-5, rtol=1e-5, check=True):
    """Drive `backend` through the scenario; returns the max abs errors per quantity."""
    N, T = int(npz["N"]), int(npz["T"])
    has_box = str(npz["task"]) == "BlindGrasping"
    assert int(npz["extra_in_reset"]) == 0
    err = {}

    def cmp(name, got, want, a=atol, r=rtol):
        got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
        e = float(np.max(np.abs(got - want))) if got.size else 0.0
        err[name] = max(err.get(name, 0.0), e)
        if check:
            np.testing.assert_allclose(got, want, atol=a, rtol=r, err_msg=name)

    # ---- env.reset()
    backend.set("reset_flag", np.ones((1, N)))
    backend.set_reset_samples(npz["reset0_samples"])
    backend.reset_flagged_no_physics()
    inject(backend, npz, "l1reset0")
    backend.compute_observations()
    backend.l2_step_no_reset()
    cmp("obs0", backend.obs_buf(), npz["obs0"])
    assert not backend.get("reset_flag").any()

    for t in range(T):
        backend.process_actions(npz["actions"][t])
        inject(backend, npz, "l1", t)
        backend.l2_step_no_reset()
        cmp(f"obs", backend.obs_buf(), npz["obs"][t])
        cmp(f"rew", backend.get("rew")[0], npz["rew"][t], a=2e-3, r=2e-6)
        done = backend.get("reset_flag")[0].astype(bool)
        if check:
            assert (done == npz["done"][t]).all(), f"done mismatch at step {t}"
        cmp("rew_total", backend.get("rew_comp")[52], npz["rew_total"][t], a=2e-3, r=2e-6)
        if done.any():
            assert bool(npz["reset_l1_used"][t])
            backend.set_reset_samples(npz["reset_samples"][t])
            backend.reset_flagged_no_physics()
            inject(backend, npz, "l1r", t)
        # everything below was recorded after env.step() returned, i.e. after the in-step resets
        if has_box:
            ts = npz["task_state"][t]
            cmp("current_stage", backend.get("current_stage")[0], ts[0])
            cmp("time_in_stage", backend.get("time_in_stage")[0], ts[1], a=1e-5)
            cmp("stage_contact_duration", backend.get("stage_contact_duration")[0], ts[2], a=1e-5)
            cmp("success_duration_steps", backend.get("success_duration_steps")[0], ts[3])
            cmp("just2", backend.get("just2")[0], ts[4])
            cmp("just3", backend.get("just3")[0], ts[5])
        cmp("targets", backend.get("targets").T, npz["targets"][t])
        cmp("active_prev_targets", backend.get("active_prev_targets").T, npz["active_prev_targets"][t])
        cmp("episode_step", backend.get("episode_step")[0], npz["episode_step"][t])
    return err
