"""Replay of the golden L2 scenarios (tests/golden/l2_*.npz, produced by the reference's own Python) against
a backend exposing the oracle-style stage interface.  Used with the CPU oracle (-m "not gpu") and with the
HIP kernels through the C-ABI (-m gpu)."""
import json

import numpy as np

from dexrobot_isaac_amd.config import OBS_KEYS, REWARD_TERMS, build_sim_config, default_cfg, obs_key_offsets

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


ALL_SCENARIOS = ["blind_default", "blind_fast", "base_default", "base_position", "blind_wide", "base_wide", "blind_long"]


def check_reward_table(rc, names, values, cmp, tag):
    """rc: the backend's rew_comp rows [59][N]; names/values: the reference's reward_components dict of one step."""
    for name, want in zip(names, values):
        if np.isnan(want).all():
            continue                                        # key absent from the reference's dict at that step
        if name == "total":
            cmp(f"{tag}:total", rc[52], want, a=2e-3, r=2e-6)
        elif name.startswith("termination_"):
            base = name[len("termination_"):]
            w = base.endswith("_weighted")
            j = ("success", "failure_penalty", "timeout_penalty").index(base[:-len("_weighted")] if w else base)
            cmp(f"{tag}:{name}", rc[(56 if w else 53) + j], want, a=1e-4 if w else 0.0, r=0.0)
        else:
            w = name.endswith("_weighted")
            i = REWARD_TERMS.index(name[:-len("_weighted")] if w else name)
            cmp(f"{tag}:{name}", rc[(26 if w else 0) + i], want, a=2e-3 if w else 1e-5, r=2e-6 if w else 1e-5)


def check_finals(backend, npz, cmp):
    """End-of-scenario snapshot: obs_dict components that are not part of obs_buf + every reward component."""
    offs = obs_key_offsets()
    oa = backend.get("obs_all")
    for key, _ in OBS_KEYS:
        f = f"final_{key}"
        if f in npz.files:
            off, dim = offs[key]
            cmp(f"final:{key}", oa[off:off + dim].T, npz[f])
    names = [n[len("final_rc_"):] for n in npz.files if n.startswith("final_rc_")]
    check_reward_table(backend.get("rew_comp"), names, [npz[f"final_rc_{n}"] for n in names], cmp, "final_rc")


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

    events = {int(t): i for i, t in enumerate(npz["ev_step"])} if "ev_step" in npz.files else {}
    snaps = {int(t): i for i, t in enumerate(npz["snap_step"])} if "snap_step" in npz.files else {}
    snap_keys = json.loads(str(npz["snap_obs_keys"])) if snaps else []
    snap_rc_names = json.loads(str(npz["snap_rc_names"])) if snaps else []
    offs = obs_key_offsets()
    for t in range(T):
        if t in events:                                     # explicit env.reset_idx(ids) between two steps
            i = events[t]
            ids = npz["ev_ids"][i]
            ids = ids[ids >= 0]
            backend.set_reset_samples(npz["ev_samples"][i])
            backend.reset_idx(ids)                          # the product's reset_idx entry point (incl. its physics step)
            inject(backend, npz, "ev_l1", i)               # ... whose result is replaced by the scripted L1 state
            cmp("ev:targets", backend.get("targets").T, npz["ev_targets"][i])
            cmp("ev:active_prev_targets", backend.get("active_prev_targets").T, npz["ev_active_prev_targets"][i])
            cmp("ev:episode_step", backend.get("episode_step")[0], npz["ev_episode_step"][i])
            cmp("ev:prev_actions", backend.get("prev_actions").T[:, :npz["ev_prev_actions_obs"].shape[2]], npz["ev_prev_actions_obs"][i])
            assert (backend.get("prev_dof_pos")[:, ids] == 0).all() and (backend.get("contact_duration_steps")[:, ids] == 0).all()
            if has_box:
                ts = npz["ev_task_state"][i]
                for j, f in enumerate(("current_stage", "time_in_stage", "stage_contact_duration", "success_duration_steps", "just2", "just3")):
                    cmp(f"ev:{f}", backend.get(f)[0], ts[j], a=1e-5)
        backend.process_actions(npz["actions"][t])
        inject(backend, npz, "l1", t)
        backend.l2_step_no_reset()
        cmp(f"obs", backend.obs_buf(), npz["obs"][t])
        cmp(f"rew", backend.get("rew")[0], npz["rew"][t], a=2e-3, r=2e-6)
        done = backend.get("reset_flag")[0].astype(bool)
        if check:
            assert (done == npz["done"][t]).all(), f"done mismatch at step {t}"
        cmp("rew_total", backend.get("rew_comp")[52], npz["rew_total"][t], a=2e-3, r=2e-6)
        if "stats" in npz.files:                            # success / failure / timeout rate, consecutive successes
            cmp("stats", backend.stats()[[12, 13, 14, 15]], npz["stats"][t][:4], a=1e-6)
        if t in snaps:                                      # every obs_dict key and every reward component of this step
            oa = backend.get("obs_all")
            want, col = npz["snap_obs_all"][snaps[t]], 0
            for key in snap_keys:
                off, dim = offs[key]
                dim = want.shape[1] - col if key == snap_keys[-1] else min(dim, want.shape[1] - col)
                if key == "prev_actions":
                    dim = int(npz["actions"].shape[2])
                cmp(f"snap:{key}", oa[off:off + dim].T, want[:, col:col + dim])
                col += dim
            assert col == want.shape[1]
            check_reward_table(backend.get("rew_comp"), snap_rc_names, npz["snap_rc"][snaps[t]], cmp, "snap_rc")
        if done.any():
            assert bool(npz["reset_l1_used"][t])
            backend.set_reset_samples(npz["reset_samples"][t])
            backend.reset_flagged_no_physics()
            if "l1r_index" in npz.files:
                inject(backend, npz, "l1r", int(npz["l1r_index"][t]))
            else:
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
    check_finals(backend, npz, cmp)
    return err
