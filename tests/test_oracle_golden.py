"""The CPU oracle's L2 restatement against golden vectors produced by the reference's own Python
(tests/golden/generate_golden.py).  This is what PINS the oracle (prompt ③)."""
import os

import numpy as np
import pytest

from dexrobot_isaac_amd.config import OBS_KEYS, REWARD_TERMS, build_sim_config, obs_key_offsets
from oracle.oracle import Oracle
from tests.l2_replay import replay, scenario_config

SCENARIOS = ["blind_default", "blind_fast", "base_default", "base_position"]


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, f"l2_{name}.npz"), allow_pickle=False)


@pytest.mark.parametrize("name", SCENARIOS)
def test_oracle_replays_reference_l2(golden_dir, name):
    npz = _load(golden_dir, name)
    sc, model = build_sim_config(scenario_config(npz))
    o = Oracle(sc, model.to_struct())
    err = replay(o, npz)
    assert err["obs"] < 2e-5
    # final obs_dict components that are not part of obs_buf
    offs = obs_key_offsets()
    oa = o.get("obs_all")
    for key in ("hand_pose_arr_aligned", "fingertip_poses_hand", "fingerpad_poses_hand", "contact_forces",
                "contact_force_magnitude", "all_finger_dof_vel", "all_finger_dof_target", "active_rule_targets",
                "finger_to_object_distances", "finger_to_object_height_diff", "hand_to_object_distance",
                "grasp_state", "grasp_duration", "thumb_contact", "other_fingers_contact"):
        f = f"final_{key}"
        if f in npz.files:
            off, dim = offs[key]
            np.testing.assert_allclose(oa[off:off + dim].T, npz[f], atol=2e-5, rtol=1e-5, err_msg=key)
    # reward components of the last step
    rc = o.get("rew_comp")
    for i, term in enumerate(REWARD_TERMS):
        f = f"final_rc_{term}"
        if f in npz.files:
            np.testing.assert_allclose(rc[i], npz[f], atol=1e-5, rtol=1e-5, err_msg=term)
            np.testing.assert_allclose(rc[26 + i], npz[f"final_rc_{term}_weighted"], atol=2e-3, rtol=2e-6)
    for j, nme in enumerate(("success", "failure_penalty", "timeout_penalty")):
        np.testing.assert_allclose(rc[53 + j], npz[f"final_rc_termination_{nme}"], atol=0)
        np.testing.assert_allclose(rc[56 + j], npz[f"final_rc_termination_{nme}_weighted"], atol=1e-4)


def test_scenarios_cover_the_branches(golden_dir):
    """The fast scenario must actually exercise transitions, success, every failure type and timeouts."""
    npz = _load(golden_dir, "blind_fast")
    ts = npz["task_state"]
    assert (ts[:, 0] == 2).any() and (ts[:, 0] == 3).any()
    assert ts[:, 4].any() and ts[:, 5].any()
    assert npz["done"].sum() > 50
    assert (npz["rew"] > 1500).any()        # termination_success bonus paid
    assert (npz["rew"] < -150).any()        # failure / timeout penalty paid
    assert (npz["stats"][:, 0] > 0).any() and (npz["stats"][:, 1] > 0).any()
    base = _load(golden_dir, "base_default")
    assert base["done"].sum() == 12          # timeouts at episodeLength - 1
