"""N > 1 path on CPU: two ranks (gloo, 127.0.0.1) each own a shard of envs, step independently (no collective
inside step) and gather the rollout into the PPO buffer with one all_gather per tensor.  On the GPU
box the same code runs over RCCL/xGMI (backend "nccl")."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dexrobot_isaac_amd import default_cfg, make_env
    from dexrobot_isaac_amd.rollout import RolloutBuffer
    from oracle.py_backend import OracleCore
    N, T = 6, 4
    cfg = default_cfg("BlindGrasping")
    cfg["train"]["seed"] = 42 + rank                      # rank-local reset stream
    env = make_env("BlindGrasping", N, "cpu", "cpu", 0, cfg=cfg, _core_factory=OracleCore)
    env.reset()
    buf = RolloutBuffer(T, N, env.num_observations, "cpu", mode=mode)
    g = torch.Generator().manual_seed(1234 + rank)
    local = []
    for t in range(T):
        obs, rew, done, _ = env.step(2 * torch.rand(N, 18, generator=g) - 1)
        buf.add(obs, rew, done)
        local.append((obs.clone(), rew.clone(), done.clone()))
    assert buf.full()
    handle = buf.gather_async()                # collectives in flight; the env keeps stepping into the other slot
    assert buf.t == 0
    for t in range(2):
        obs, rew, done, _ = env.step(2 * torch.rand(N, 18, generator=g) - 1)
        buf.add(obs, rew, done)
    res = handle()
    if mode == "learner" and rank != 0:                   # non-learner ranks only send
        assert res is None
        np.save(os.path.join(out_dir, f"local_{rank}.npy"), torch.stack([o for o, _, _ in local]).numpy())
        dist.barrier()
        dist.destroy_process_group()
        return
    obs_g, rew_g, done_g = res
    assert obs_g.shape == (T, world * N, env.num_observations) and rew_g.shape == (T, world * N)
    for t in range(T):                                    # env index = rank * N + local index
        assert torch.equal(obs_g[t, rank * N:(rank + 1) * N], local[t][0])
        assert torch.equal(rew_g[t, rank * N:(rank + 1) * N], local[t][1])
        assert torch.equal(done_g[t, rank * N:(rank + 1) * N].bool(), local[t][2])
    np.save(os.path.join(out_dir, f"obs_{rank}.npy"), obs_g.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_all_gather(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), "all"), nprocs=world, join=True)
    a, b = np.load(tmp_path / "obs_0.npy"), np.load(tmp_path / "obs_1.npy")
    assert np.array_equal(a, b)                           # every rank holds the same gathered PPO buffer
    assert not np.array_equal(a[:, :6], a[:, 6:])         # shards are different envs (different seeds/actions)


def test_two_rank_shard_and_gather_to_learner(tmp_path):
    """mode="learner": only rank 0 receives (north_star: gather into the PPO buffer); rank 1 only sends."""
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), "learner"), nprocs=world, join=True)
    a = np.load(tmp_path / "obs_0.npy")
    assert not (tmp_path / "obs_1.npy").exists()
    assert np.array_equal(a[:, 6:], np.load(tmp_path / "local_1.npy"))   # rank 1's shard arrived at the learner


def test_single_process_gather_is_identity():
    from dexrobot_isaac_amd.rollout import RolloutBuffer
    buf = RolloutBuffer(2, 3, 5, "cpu")
    buf.add(torch.ones(3, 5), torch.ones(3), torch.zeros(3, dtype=torch.bool))
    buf.add(2 * torch.ones(3, 5), torch.ones(3), torch.ones(3, dtype=torch.bool))
    o, r, d = buf.gather()
    assert o.shape == (2, 3, 5) and float(o[1, 0, 0]) == 2 and d[1].all() and buf.t == 0
    import pytest
    buf.add(torch.ones(3, 5), torch.ones(3), torch.zeros(3, dtype=torch.bool))
    buf.add(torch.ones(3, 5), torch.ones(3), torch.zeros(3, dtype=torch.bool))
    with pytest.raises(RuntimeError, match="full"):
        buf.add(torch.ones(3, 5), torch.ones(3), torch.zeros(3, dtype=torch.bool))
