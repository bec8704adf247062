"""dexrobot_isaac_amd: MI355X-native vectorised DexHand simulator behind the dexhand_env env surface.

    from dexrobot_isaac_amd import make_env
    env = make_env("BlindGrasping", num_envs=4096, sim_device="cuda:0", rl_device="cuda:0", graphics_device_id=0)
    obs = env.reset(); obs, rew, done, info = env.step(actions)
"""
from .config import default_cfg  # noqa: F401
from .factory import create_dex_env, make_env  # noqa: F401
