"""World-size-2 gloo tests (CPU) of the N>1 path: the flat-bucket gradient all-reduce of the shared
learner, and the env-shard seeding rule (the fused HIP step under a process group: tests/test_multigpu_paths_gpu.py)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from distributional_rl_navigation_amd.iqn.agent import IQNAgent


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _batch(seed, B=32):
    g = np.random.RandomState(seed)
    return (torch.from_numpy(g.normal(0, 5, (B, 26)).astype(np.float32)),
            torch.from_numpy(g.randint(9, size=(B, 1)).astype(np.int64)),
            torch.from_numpy(g.normal(0, 3, (B, 1)).astype(np.float32)),
            torch.from_numpy(g.normal(0, 5, (B, 26)).astype(np.float32)),
            torch.from_numpy((g.uniform(size=(B, 1)) < 0.2).astype(np.float32)))


def _taus(seed, B=32):
    g = np.random.RandomState(1000 + seed)
    return torch.from_numpy(g.uniform(size=(B, 8)).astype(np.float32)), torch.from_numpy(g.uniform(size=(B, 8)).astype(np.float32))


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    agent = IQNAgent(26, 9, BATCH_SIZE=32, BUFFER_SIZE=64, seed=3, distributed=True)
    for step in range(3):
        tt, tl = _taus(10 * step + rank)
        agent.train(_batch(10 * step + rank), taus_target=tt, taus_local=tl)
    flat = torch.cat([p.detach().reshape(-1) for p in agent.qnetwork_local.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        torch.save(dict(params=flat, same=bool(torch.equal(gathered[0], gathered[1]))), out)
    dist.destroy_process_group()


def test_shared_learner_allreduce_equals_big_batch(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = torch.load(out)
    assert res["same"], "ranks diverged: all-reduced gradients must keep shared learners identical"
    # single-process equivalent: loss = mean over the union of both ranks' batches = mean of the two
    # per-rank mean losses, so averaged per-rank gradients == gradients of the 64-row batch
    ref = IQNAgent(26, 9, BATCH_SIZE=64, BUFFER_SIZE=64, seed=3)
    for step in range(3):
        b0, b1 = _batch(10 * step), _batch(10 * step + 1)
        (t0, l0), (t1, l1) = _taus(10 * step), _taus(10 * step + 1)
        exp = tuple(torch.cat([a, b]) for a, b in zip(b0, b1))
        ref.train(exp, taus_target=torch.cat([t0, t1]), taus_local=torch.cat([l0, l1]))
    flat = torch.cat([p.detach().reshape(-1) for p in ref.qnetwork_local.parameters()])
    np.testing.assert_allclose(res["params"].numpy(), flat.numpy(), rtol=0, atol=2e-6)


def test_shard_seeds_tile_the_single_gpu_run():
    """`shard_seeds` (the function VecMarineNavEnv seeds its envs with): rank r of `world` ranks with
    first_index = r * n gets exactly rows [r n, (r+1) n) of the one-GPU seed vector -- for BASELINE configs[3]'s
    524 288 = 8 x 65 536 split, for a ragged last shard, and across the 2^32 wrap of RandomState's seed range."""
    import pytest
    from distributional_rl_navigation_amd.marinenav_env.vec_env import shard_seeds
    n, world = 65536, 8
    full = shard_seeds(n * world, seed=0)
    assert full.dtype == np.uint32 and full[0] == 0 and full[-1] == n * world - 1
    for r in range(world):
        assert np.array_equal(shard_seeds(n, seed=0, first_index=r * n), full[r * n:(r + 1) * n])
    # base seed offsets every env; shards still tile
    full = shard_seeds(1000, seed=348)
    parts = [shard_seeds(m, seed=348, first_index=f) for f, m in ((0, 300), (300, 300), (600, 400))]
    assert np.array_equal(np.concatenate(parts), full) and full[0] == 348
    # wrap-around at 2^32
    w = shard_seeds(8, seed=(1 << 32) - 3)
    assert list(w) == [4294967293, 4294967294, 4294967295, 0, 1, 2, 3, 4]
    assert np.array_equal(shard_seeds(4, seed=(1 << 32) - 3, first_index=4), w[4:])
    with pytest.raises(ValueError):
        shard_seeds(0)
    with pytest.raises(ValueError):
        shard_seeds(4, first_index=-1)


def test_pool_worker_device_selection():
    """train_iqn -P: `-D cuda:K` pins every worker, `-D cuda` / no -D spreads worker i to GPU i modulo the visible GPUs, `-D cpu` passes
    through (set_device was called unconditionally and every worker landed on one device)."""
    from distributional_rl_navigation_amd.train_iqn import _worker_device
    assert [_worker_device(None, i, 4) for i in range(6)] == ["cuda:0", "cuda:1", "cuda:2", "cuda:3", "cuda:0", "cuda:1"]
    assert [_worker_device("cuda", i, 2) for i in range(3)] == ["cuda:0", "cuda:1", "cuda:0"]
    assert _worker_device("cuda:3", 5, 8) == "cuda:3" and _worker_device("cpu", 1, 0) == "cpu" and _worker_device(None, 2, 0) == "cuda:0"
