"""The CPU oracle's L2 restatement against golden vectors produced by the reference's own Python
(tests/golden/generate_golden.py).  This is what PINS the oracle (prompt ③)."""
import json
import os

import numpy as np
import pytest

from dexrobot_isaac_amd.config import OBS_KEYS, REWARD_TERMS, build_sim_config, obs_key_offsets
from oracle.oracle import Oracle
from tests.l2_replay import replay, scenario_config

from tests.l2_replay import ALL_SCENARIOS as SCENARIOS  # noqa: E402


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, f"l2_{name}.npz"), allow_pickle=False)


@pytest.mark.parametrize("name", SCENARIOS)
def test_oracle_replays_reference_l2(golden_dir, name):
    npz = _load(golden_dir, name)
    sc, model = build_sim_config(scenario_config(npz))
    o = Oracle(sc, model.to_struct())
    err = replay(o, npz)
    assert err["obs"] < 2e-5
    # (replay also checks, on this backend: the end-of-scenario snapshot of the obs_dict components outside obs_buf and of
    # every reward component, the per-step success / failure / timeout rates and consecutive successes, and -- in the
    # round-2 scenarios -- explicit reset_idx(ids) events and per-step snapshots of every obs_dict key / reward component)
    assert any(k.startswith("final:") for k in err) and any(k.startswith("final_rc:") for k in err)
    if name in ("blind_wide", "base_wide"):
        assert any(k.startswith("ev:") for k in err) and any(k.startswith("snap:") for k in err)


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
    # round 2: default-length FSM (stage 1 = 200 control steps) reaching stage 2, stage 3 and grasp_lift_success
    lng = _load(golden_dir, "blind_long")
    ts = lng["task_state"]
    assert (ts[:, 0] == 2).sum() > 20 and (ts[:, 0] == 3).sum() > 50 and (lng["rew"] > 1500).any()
    assert json.loads(str(lng["cfg_overrides"])) == {}
    # ... and two workgroups incl. a padded one, with explicit reset_idx events
    wide = _load(golden_dir, "blind_wide")
    assert int(wide["N"]) == 70 and len(wide["ev_step"]) == 2 and wide["done"].sum() > 200
