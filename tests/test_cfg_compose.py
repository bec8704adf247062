"""The Hydra-free composer against (a) a miniature cfg tree using every feature and (b), when the reference checkout
is present, the reference's real YAML tree -- whose composition must equal the literal defaults the product ships."""
import os

import pytest

from dexrobot_isaac_amd.cfg_compose import compose
from dexrobot_isaac_amd.config import build_sim_config, default_cfg

REF_CFG = "/root/reference/dexhand_env/cfg"


def _write(root, rel, text):
    p = os.path.join(root, rel)
    os.makedirs(os.path.dirname(p), exist_ok=True)
    with open(p, "w") as f:
        f.write(text)


def test_composer_semantics(tmp_path):
    d = str(tmp_path)
    _write(d, "config.yaml", "defaults:\n  - train: ppo\n  - base/video\n  - _self_\n  - task: A\nsim:\n  dt: 0.005\nenv:\n  numEnvs: 16\n")
    _write(d, "train/ppo.yaml", "# @package train\nlr: 0.1\n")
    _write(d, "base/video.yaml", "# @package _global_\nenv:\n  videoCodec: mp4v\n")
    _write(d, "physics/default.yaml", "# @package sim\nsubsteps: 4\nphysx:\n  iters: 16\n")
    _write(d, "physics/fast.yaml", "# @package sim\ndefaults:\n  - default\n  - _self_\nsubsteps: 2\n")
    _write(d, "task/A.yaml", "# @package _global_\ndefaults:\n  - /physics/default\n  - _self_\nsim:\n  dt: 0.01\ntask:\n  name: A\n  w:\n    a: 1\n")
    _write(d, "task/B.yaml", "# @package _global_\ndefaults:\n  - A\n  - _self_\nenv:\n  numEnvs: ${env.numEnvs}\n  half: ${task.w.a}\ntask:\n  name: B\n  w:\n    _delete_: true\n    b: 2\n")
    c = compose(d)
    assert c["train"] == {"lr": 0.1} and c["env"] == {"videoCodec": "mp4v", "numEnvs": 16}
    assert c["sim"] == {"substeps": 4, "physx": {"iters": 16}, "dt": 0.01}          # task merged after _self_
    c = compose(d, overrides=["task=B", "env.numEnvs=64", "+extra.k=[1, 2]"])
    assert c["task"]["name"] == "B" and c["task"]["w"] == {"a": 1, "_delete_": True, "b": 2}
    assert c["env"]["numEnvs"] == 64 and c["env"]["half"] == 1 and c["extra"]["k"] == [1, 2]
    assert compose(d, overrides=["task=B"])["env"]["numEnvs"] == 16                  # self-interpolation keeps 16
    with pytest.raises(FileNotFoundError):
        compose(d, overrides=["task=nope"])
    _write(d, "task/C.yaml", "# @package _global_\nenv:\n  x: ${env.missing}\n")
    with pytest.raises(ValueError, match="not found"):
        compose(d, overrides=["task=C"])


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="reference checkout not present")
@pytest.mark.parametrize("task", ["BaseTask", "BlindGrasping"])
def test_reference_yaml_tree_equals_shipped_defaults(task):
    composed = compose(REF_CFG, overrides=[f"task={task}", "env.numEnvs=4096"])
    shipped = default_cfg(task)
    shipped["env"]["numEnvs"] = 4096
    a, _ = build_sim_config(composed)
    b, _ = build_sim_config(shipped)
    assert bytes(a) == bytes(b)                    # every number the engine consumes is identical
    for sect in ("task", "sim"):
        for k, v in shipped[sect].items():
            if k in ("reward_weights", "physics_engine", "graphicsDeviceId"):
                continue
            assert composed[sect][k] == v, (sect, k)
    rw = {k: v for k, v in composed["task"]["reward_weights"].items() if k != "_delete_"}
    assert rw == shipped["task"]["reward_weights"]
