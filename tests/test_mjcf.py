"""MJCF loader.  No reference MJCF exists offline (absent submodule), so the loader is pinned by a round
trip through its own exporter and by driving the oracle with the re-loaded model."""
import numpy as np
import pytest

from dexrobot_isaac_amd.config import build_sim_config, default_cfg
from dexrobot_isaac_amd.hand_model import HandModel
from dexrobot_isaac_amd.mjcf import export_mjcf, load_mjcf


def test_round_trip_reproduces_the_model():
    ref = HandModel()
    xml = export_mjcf(ref)
    assert "r_f_joint5_4" in xml and 'kp="10000"' in xml and "right_hand_base" in xml
    got = load_mjcf(xml)
    for name in ("jtype", "jpoff", "jaxis", "mass", "com", "inertia", "kp", "kd", "armature", "lo", "hi", "jRoff",
                 "site_parent", "site_p", "site_R", "cap_parent", "cap_p0", "cap_p1", "cap_r", "cap_fslot", "spawn_pos",
                 "spawn_rot", "body_parent", "body_p", "body_R", "body_fslot"):
        np.testing.assert_allclose(getattr(got, name), getattr(ref, name), atol=2e-7, err_msg=name)
    a, b = ref.to_struct(), got.to_struct()
    assert np.allclose(np.frombuffer(bytes(a), dtype=np.float32)[:200], np.frombuffer(bytes(b), dtype=np.float32)[:200], atol=1e-6)


def test_loaded_model_drives_the_oracle_identically():
    from oracle.oracle import Oracle
    cfg = default_cfg("BlindGrasping")
    cfg["env"]["numEnvs"] = 2
    sc, ref = build_sim_config(cfg)
    got = load_mjcf(export_mjcf(ref))
    o1, o2 = Oracle(sc, ref.to_struct()), Oracle(sc, got.to_struct())
    o1.reset(), o2.reset()
    rng = np.random.default_rng(0)
    for _ in range(5):
        a = (2 * rng.random((2, 18)) - 1).astype(np.float32)
        ob1, _, _ = o1.step(a)
        ob2, _, _ = o2.step(a)
    np.testing.assert_allclose(ob2, ob1, atol=5e-6)


def test_wrong_topology_is_rejected():
    xml = export_mjcf(HandModel()).replace('name="r_f_joint3_1"', 'name="some_other_joint"')
    with pytest.raises(ValueError, match="unexpected joint|lacks DexHand joints"):
        load_mjcf(xml)
