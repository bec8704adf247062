"""Factory with the reference's signature (dexhand_env/factory.py:20-147)."""
import copy

from .config import default_cfg
from .env import DexHandEnv


def create_dex_env(task_name, cfg, rl_device, sim_device, graphics_device_id, force_render=False, video_config=None,
                   **kwargs):
    if task_name not in ("BaseTask", "BlindGrasping"):
        raise ValueError(f"Unknown task: {task_name}")                       # factory.py:61-62
    headless = not cfg["env"]["viewer"]                                      # factory.py:67
    if not headless:
        raise RuntimeError("env.viewer=true: the Isaac Gym viewer is out of scope of this engine (headless only)")
    return DexHandEnv(cfg, task_name, rl_device, sim_device, graphics_device_id, headless, force_render, video_config,
                      **kwargs)


def make_env(task_name: str, num_envs: int, sim_device: str, rl_device: str, graphics_device_id: int = 0,
             cfg: dict = None, force_render: bool = False, video_config: dict = None, **kwargs):
    """make_env(task_name, num_envs, sim_device, rl_device, graphics_device_id, cfg=None, ...) -> env."""
    if cfg is None:
        cfg = default_cfg(task_name)
    else:
        cfg = copy.deepcopy(cfg)
    cfg["env"]["numEnvs"] = num_envs                                         # factory.py:129-134
    if cfg["task"].get("name", task_name) != task_name:
        raise ValueError(f"cfg is for task '{cfg['task'].get('name')}' but '{task_name}' was requested")
    return create_dex_env(task_name, cfg, rl_device, sim_device, graphics_device_id, force_render, video_config, **kwargs)
