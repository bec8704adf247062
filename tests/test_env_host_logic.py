"""Host logic of the product (DexHandEnv / factory / spaces) exercised on CPU by injecting the oracle as the
engine (test infrastructure; the product itself has no CPU path).  Mirrors what the reference's smoke harness
checks (examples/dexhand_test.py:1707-1782: coupling invariants, shapes) plus the drop-in surface of SURVEY §8b."""
import numpy as np
import pytest
import torch

from dexrobot_isaac_amd import default_cfg, make_env
from dexrobot_isaac_amd._lib import DexSimError
from oracle.py_backend import OracleCore


def _env(task, n, **kw):
    return make_env(task, n, "cpu", "cpu", 0, _core_factory=OracleCore, **kw)


def test_factory_contract():
    with pytest.raises(ValueError, match="Unknown task"):
        make_env("NoSuchTask", 4, "cuda:0", "cuda:0", 0)
    cfg = default_cfg("BaseTask")
    cfg["sim"]["use_gpu_pipeline"] = True
    with pytest.raises(RuntimeError, match="deprecated"):
        make_env("BaseTask", 4, "cpu", "cpu", 0, cfg=cfg, _core_factory=OracleCore)
    with pytest.raises(DexSimError, match="no CPU fallback|HIP"):
        make_env("BaseTask", 4, "cpu", "cpu", 0)          # product path: no CPU engine, fails loudly
    cfg = default_cfg("BlindGrasping")
    cfg["task"]["termination"]["active_failure_criteria"] = ["no_such_criterion"]
    with pytest.raises(RuntimeError, match="not implemented"):
        make_env("BlindGrasping", 4, "cpu", "cpu", 0, cfg=cfg, _core_factory=OracleCore)
    cfg = default_cfg("BlindGrasping")
    cfg["task"]["policy_observation_keys"].append("bogus_key")
    with pytest.raises(RuntimeError, match="MISSING"):
        make_env("BlindGrasping", 4, "cpu", "cpu", 0, cfg=cfg, _core_factory=OracleCore)


@pytest.mark.parametrize("task,nobs", [("BaseTask", 224), ("BlindGrasping", 158)])
def test_config1_plumbing_200_random_steps(task, nobs):
    """BASELINE configs[0]: num_envs=4, random actions, 200 steps (here both tasks)."""
    env = _env(task, 4)
    assert (env.num_envs, env.num_observations, env.num_actions) == (4, nobs, 18)
    assert env.observation_space.shape == (nobs,) and env.action_space.shape == (18,)
    assert env.action_space.low.min() == -1 and env.action_space.high.max() == 1
    assert env.get_env_info()["num_envs"] == 4 and env.get_number_of_agents() == 1
    assert abs(env.physics_manager.control_dt - 2 * env.physics_manager.physics_dt) < 1e-9
    obs = env.reset()
    assert obs.shape == (4, nobs) and obs.dtype == torch.float32
    assert (env.episode_step_count == 1).all()                  # reset() runs post_physics_step once (dexhand_base.py:837)
    g = torch.Generator().manual_seed(0)
    dones = 0
    for t in range(200):
        a = 2 * torch.rand(4, 18, generator=g) - 1
        obs, rew, done, info = env.step(a)
        assert obs.shape == (4, nobs) and rew.shape == (4,) and done.dtype == torch.bool
        assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
        dones += int(done.sum())
        # coupling invariants of the reference harness (examples/dexhand_test.py:1707-1758)
        tg = env.full_dof_targets
        live = ~done                                             # reset envs carry the raw randomised pose as target
        assert torch.allclose(tg[live, 8], tg[live, 9]) and torch.allclose(tg[live, 18], tg[live, 10])
        assert torch.allclose(tg[live, 22], 2 * tg[live, 10]) and (tg[live, 14] == 0).all()
    assert {"consecutive_successes", "episode_length", "success", "failure", "timeout", "reward_components",
            "success_rate", "failure_rate", "timeout_rate"} <= set(info)
    rc = info["reward_components"]
    assert "total" in rc and all(v.shape == (4,) for v in rc.values())
    assert info["success"].dtype == torch.bool
    if task == "BlindGrasping":
        assert dones >= 4          # stage-1 quality check fails at t = 4 s for a random policy
        assert {"s1_height_alignment", "s1_height_alignment_weighted", "termination_failure_penalty"} <= set(rc)
        assert "alive" not in rc                                 # zero-weight terms are not reported
    else:
        assert "alive" in rc and "s1_completion" not in rc


def test_obs_dict_views_and_accessors():
    env = _env("BlindGrasping", 3)
    env.reset()
    env.step(torch.zeros(3, 18))
    od = env.get_observations_dict()
    assert od["base_dof_pos"].shape == (3, 6) and od["fingerpad_distances"].shape == (3, 10)
    assert od["current_stage"].shape == (3, 1) and (od["current_stage"] == 1).all()
    # obs_buf is the concatenation of the configured keys in order (observation_encoder.py:783-829)
    cat = torch.cat([od[k] for k in env.task_cfg["policy_observation_keys"]], dim=1)
    assert torch.allclose(cat, env.obs_buf)
    lo, hi = env.observation_encoder.component_slice_indices["hand_pose"]
    assert torch.allclose(env.obs_buf[:, lo:hi], od["hand_pose"])
    th = env.observation_encoder.get_raw_finger_dof("r_f_joint1_1", "pos", od)
    assert torch.allclose(th, env.dof_pos[:, 6], atol=1e-6)
    # hand_pose quaternion at ARR = 0 would be [0, sqrt.5, 0, sqrt.5]; arr-aligned compensates it
    assert env.dof_props.shape == (26, 6) and float(env.dof_props[0, 0]) == 10000.0 and float(env.dof_props[6, 1]) == 1.0
    assert env.action_processor.max_deltas.shape == (18,)
    assert abs(float(env.action_processor.max_deltas[0]) - 0.02 * 0.1) < 1e-9
    ua = env.action_processor.unscale_actions(torch.ones(3, 18))
    assert torch.allclose(ua, env.action_processor.max_deltas.expand(3, 18))


def test_reset_idx_and_hooks():
    env = _env("BlindGrasping", 6)
    env.reset()
    for _ in range(3):
        env.step(torch.zeros(6, 18))
    before = env.episode_step_count.clone()
    env.reset_idx(torch.tensor([], dtype=torch.long))            # early return (dexhand_base.py:746-747)
    assert (env.episode_step_count == before).all()
    env.reset_idx(torch.tensor([1, 4]))
    assert env.episode_step_count[[1, 4]].tolist() == [0, 0] and (env.episode_step_count[[0, 2, 3, 5]] == 4).all()
    # pre-action rule hook: rule targets feed the next process_actions for DOFs the policy does not control
    calls = []

    def rule(prev, state):
        calls.append(sorted(state))
        out = prev.clone()
        out[:, 0] = 0.05
        return out

    env.action_processor.set_pre_action_rule(rule)
    env.step(torch.zeros(6, 18))
    assert calls and calls[0] == ["env", "obs_dict"]
    assert torch.allclose(env.get_observations_dict()["active_rule_targets"][:, 0], torch.full((6,), 0.05))
    # registering a filter does not enable it (rules.py:113-135): the fused device path stays selected
    env.action_processor.register_post_action_filter("x", lambda a, b, c: c)
    assert not env.action_processor._host_path()
    env.action_processor._enabled_post_action_filters = ["velocity_clamp", "position_clamp", "x"]
    assert env.action_processor._host_path()
    env.action_processor._enabled_post_action_filters = ["velocity_clamp", "position_clamp"]
    with pytest.raises(RuntimeError, match="None"):
        env.step(None)


def test_pre_action_rule_follows_the_reference_ordering():
    """step_processor.py:56-77: the custom pre-action rule runs between compute_observations and
    concatenate_observations, on the terminal obs_dict and the PRE-reset active_prev_targets, and its output is what the
    policy sees when active_rule_targets is a policy observation key; reset() runs it in both observation passes
    (dexhand_base.py:805-838)."""
    cfg = default_cfg("BlindGrasping")
    cfg["env"]["episodeLength"] = 4                              # time-outs -> in-step resets at step 3
    cfg["task"]["policy_observation_keys"] = list(cfg["task"]["policy_observation_keys"]) + ["active_rule_targets"]
    env = make_env("BlindGrasping", 5, "cpu", "cpu", 0, cfg=cfg, _core_factory=OracleCore)
    assert env.num_observations == 158 + 18
    seen = []

    def rule(prev, state):
        seen.append((prev.clone(), state["obs_dict"]["active_prev_targets"].clone(), state["obs_dict"]["episode_time"].clone()))
        out = prev.clone()
        out[:, 3] = 0.25 + 0.01 * len(seen)
        return out

    env.action_processor.set_pre_action_rule(rule)
    env.reset()
    assert len(seen) == 2                                        # both observation passes of reset()
    sl = env.observation_encoder.component_slice_indices["active_rule_targets"]
    assert torch.allclose(env.obs_buf[:, sl[0] + 3], torch.full((5,), 0.27))
    done_seen = False
    for t in range(5):
        obs, rew, done, _ = env.step(0.5 * torch.ones(5, 18))
        prev, ob_prev, _ = seen[-1]
        assert torch.equal(prev, ob_prev)                        # the rule got the pre-reset targets (the obs_dict entry)
        assert torch.allclose(obs[:, sl[0] + 3], torch.full((5,), 0.25 + 0.01 * len(seen)))   # ... and the policy sees its output
        assert torch.allclose(env.action_processor.active_rule_targets[:, 3], obs[:, sl[0] + 3])
        if bool(done.any()):
            done_seen = True
            # after an in-step reset the field holds the post-reset targets, the rule was still given the terminal ones
            assert not torch.equal(env.action_processor.active_prev_targets, ob_prev)
    assert done_seen
