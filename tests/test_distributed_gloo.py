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


def _sink_worker(rank, world, port, out_dir, mode):
    """The path bench.py runs on N > 1 GPUs: RolloutBuffer.sink() before every step (the step writes its outputs and its
    statistics block straight into the slot), gather_async() when the slot is full while the env keeps stepping into the other
    slot, handles consumed one rollout late."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dexrobot_isaac_amd import _abi, default_cfg, make_env
    from dexrobot_isaac_amd.rollout import RolloutBuffer
    from oracle.py_backend import OracleCore
    N, T, ROLLOUTS = 6, 4, 3
    cfg = default_cfg("BlindGrasping")
    cfg["train"]["seed"] = 42 + rank
    cfg["env"]["episodeLength"] = 4 + rank                   # rank 0 times out every 3 steps, rank 1 every 4: the ranks' rates differ
    env = make_env("BlindGrasping", N, "cpu", "cpu", 0, cfg=cfg, _core_factory=OracleCore)
    env.reset()
    core = env._core
    buf = RolloutBuffer(T, N, env.num_observations, "cpu", mode=mode)
    g = torch.Generator().manual_seed(1234 + rank)
    local, local_stats, pending, results = [], [], [], []
    for step in range(ROLLOUTS * T):
        buf.sink(core)                                       # BEFORE the step, as in bench.py
        obs, rew, done, _ = env.step(2 * torch.rand(N, 18, generator=g) - 1)
        local.append((obs.clone(), rew.clone(), done.clone()))
        local_stats.append(core.stats.clone())
        if buf.full():
            pending.append(buf.gather_async())               # collectives in flight, next rollout fills the other slot
            if len(pending) > 1:
                h = pending.pop(0)
                results.append((h(), h.stats()))
    while pending:
        h = pending.pop(0)
        results.append((h(), h.stats()))
    assert len(results) == ROLLOUTS
    S = _abi.STAT
    mine = torch.stack(local_stats).double()                 # (ROLLOUTS * T, STAT_WORDS): this rank's own statistics
    all_stats = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(all_stats, mine)
    tot = torch.stack(all_stats).sum(0)
    cs = 0
    for r, (res, st) in enumerate(results):
        sl = slice(r * T, (r + 1) * T)
        # whole-population statistics = mean of the ranks' (equal shards), counts summed
        assert torch.allclose(st["timeout_rate"], tot[sl, S["TIMEOUT_RATE"]] / world)
        assert torch.allclose(st["failure_rate"], tot[sl, S["FAILURE_RATE"]] / world)
        assert torch.allclose(st["num_resets"], tot[sl, S["NUM_RESETS"]])
        for t in range(T):
            cs = min(cs + 1, buf.max_consecutive_successes) if tot[r * T + t, S["SUCCESS_RATE"]] > 0 else 0
            assert float(st["consecutive_successes"][t]) == cs
        if res is None:
            assert mode == "learner" and rank != 0
            continue
        obs_g, rew_g, done_g = res
        assert obs_g.shape == (T, world * N, env.num_observations)
        for t in range(T):
            o, rw, d = local[r * T + t]
            assert torch.equal(obs_g[t, rank * N:(rank + 1) * N], o)
            assert torch.equal(rew_g[t, rank * N:(rank + 1) * N], rw)
            assert torch.equal(done_g[t, rank * N:(rank + 1) * N].bool(), d)
    if rank == 0:
        np.save(os.path.join(out_dir, "timeout_rate.npy"), torch.cat([st["timeout_rate"] for _, st in results]).numpy())
        np.save(os.path.join(out_dir, "rank0_timeout_rate.npy"), mine[:, S["TIMEOUT_RATE"]].numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sink_path_with_in_flight_gather_and_reduced_statistics(tmp_path):
    """the gloo tests drove add(), not the sink() + gather_async() path of bench.py; and the logging
    rates / consecutive successes were per-rank.  Both modes."""
    for mode in ("learner", "all"):
        world, port = 2, _free_port()
        mp.spawn(_sink_worker, args=(world, port, str(tmp_path), mode), nprocs=world, join=True)
        glob, r0 = np.load(tmp_path / "timeout_rate.npy"), np.load(tmp_path / "rank0_timeout_rate.npy")
        assert r0.max() == 1.0 and (glob[(r0 == 1.0)] >= 0.5).all() and (glob == 0.5).any()   # rank 0's all-env timeouts are half of the population's
        assert (glob > 0).sum() > (r0 > 0).sum()              # steps in which only rank 1 timed out show up globally too


def test_reduce_stats_consecutive_successes_single_process():
    """The global consecutive-successes counter (termination_manager.py:323-339) from per-step success rates, carried
    across rollouts and capped."""
    from dexrobot_isaac_amd import _abi
    from dexrobot_isaac_amd.rollout import RolloutBuffer
    buf = RolloutBuffer(4, 3, 5, "cpu", max_consecutive_successes=5)
    S = _abi.STAT
    seqs = []
    for flags in ([0, 1, 1, 1], [1, 1, 1, 1], [1, 0, 1, 0]):
        for f in flags:
            st = torch.zeros(_abi.STAT_USED)
            st[S["SUCCESS_RATE"]] = f / 3.0
            buf.add(torch.zeros(3, 5), torch.zeros(3), torch.zeros(3, dtype=torch.bool), stats=st)
        seqs.append(buf.gather_async().stats()["consecutive_successes"].tolist())
    assert seqs == [[0, 1, 2, 3], [4, 5, 5, 5], [5, 0, 1, 0]]
