"""CPU, world_size 2, gloo: the sharding + single packed all-gather of gym_dockauv_amd/parallel.py.  The env step is
replaced by a deterministic stand-in (the HIP kernel needs a GPU); what is checked is the multi-rank plumbing:
contiguous shard ranges, action slicing, in-place gather into each rank's slice, double-buffer ordering."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    from gym_dockauv_amd.parallel import shard_range
    for total in (0, 1, 7, 64, 4096, 524288 + 3):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _worker(rank, world, port, n_local, n_obs, steps, overlap, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from gym_dockauv_amd.parallel import ShardedStepper, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        total = world * n_local
        first, count = shard_range(total, world, rank)
        assert count == n_local
        row = n_obs + 2
        gidx = torch.arange(first, first + count, dtype=torch.float32)
        calls = []

        def step_fn(actions_local, out_local):
            # stand-in "env": row = [global env index, step, sum(actions), ...]; reward = -index; done = index odd
            t = float(len(calls))
            calls.append(t)
            out_local[:, 0] = gidx
            out_local[:, 1] = t
            out_local[:, 2] = actions_local.sum(dim=1)
            out_local[:, 3:n_obs] = 0.5
            out_local[:, n_obs] = -gidx
            out_local[:, n_obs + 1] = (gidx % 2 == 1).float()

        st = ShardedStepper(n_local, row, step_fn, "cpu", world=world, rank=rank, overlap=overlap)
        all_actions = torch.arange(total * 3, dtype=torch.float32).reshape(total, 3)   # what a single learner emits
        results = []
        for t in range(steps):
            buf = st.step(all_actions[first:first + count] + t)
            st.wait()
            results.append(buf.clone())
        q.put((rank, [r.numpy() for r in results]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
def test_two_rank_gather_equals_single_process(overlap):
    import torch.multiprocessing as mp
    world, n_local, n_obs, steps = 2, 5, 6, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 200) + (10 if overlap else 0)
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_local, n_obs, steps, overlap, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total = world * n_local
    idx = np.arange(total, dtype=np.float32)
    acts = np.arange(total * 3, dtype=np.float32).reshape(total, 3)
    for t in range(steps):
        exp = np.zeros((total, n_obs + 2), dtype=np.float32)
        exp[:, 0] = idx
        exp[:, 1] = t
        exp[:, 2] = (acts + t).sum(axis=1)
        exp[:, 3:n_obs] = 0.5
        exp[:, n_obs] = -idx
        exp[:, n_obs + 1] = idx % 2 == 1
        for r in range(world):
            np.testing.assert_array_equal(got[r][t], exp)
    from gym_dockauv_amd.parallel import ShardedStepper
    import torch
    obs, rew, done = ShardedStepper.split(torch.from_numpy(got[0][0]), n_obs)
    assert obs.shape == (total, n_obs) and rew.shape == (total,) and done.dtype == torch.bool and done.sum() == total // 2


def _worker_bf16(rank, world, port, n_local, n_obs, q):
    """gather_dtype "bf16": rows of ceil(n_obs / 2) + 2 words; the stand-in env packs bfloat16 pairs the way the kernel does"""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from gym_dockauv_amd.parallel import ShardedStepper, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        first, count = shard_range(world * n_local, world, rank)
        npair = (n_obs + 1) // 2
        wpr = npair + 2

        def step_fn(actions_local, out_local):
            obs = (torch.arange(first, first + count, dtype=torch.float32)[:, None] * 0.37 + torch.arange(n_obs)[None, :] * 1.001)
            pad = torch.zeros((count, 2 * npair), dtype=torch.bfloat16)
            pad[:, :n_obs] = obs.to(torch.bfloat16)                         # round to nearest even
            out_local[:, :npair] = pad.view(torch.float32)
            out_local[:, npair] = actions_local.sum(dim=1)
            out_local[:, npair + 1] = (torch.arange(first, first + count) % 3 == 0).float()

        st = ShardedStepper(n_local, wpr, step_fn, "cpu", world=world, rank=rank, overlap=False, gather_dtype="bf16")
        assert st.bytes_per_rank_per_step == n_local * wpr * 4
        buf = st.step(torch.ones((count, 3)) * (rank + 1))
        st.wait()
        obs, rew, done = ShardedStepper.split_bf16(buf, n_obs)
        q.put((rank, obs.float().numpy(), rew.numpy(), done.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_obs", [20, 7])
def test_two_rank_bf16_gather(n_obs):
    import torch
    import torch.multiprocessing as mp
    world, n_local = 2, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29850 + (os.getpid() % 100) + n_obs
    procs = [ctx.Process(target=_worker_bf16, args=(r, world, port, n_local, n_obs, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, obs, rew, done = q.get(timeout=120)
        got[r] = (obs, rew, done)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total = world * n_local
    exp = (torch.arange(total, dtype=torch.float32)[:, None] * 0.37 + torch.arange(n_obs)[None, :] * 1.001).to(torch.bfloat16).float().numpy()
    for r in range(world):
        obs, rew, done = got[r]
        np.testing.assert_array_equal(obs, exp)
        np.testing.assert_array_equal(rew, np.repeat([3.0, 6.0], n_local))
        np.testing.assert_array_equal(done, np.arange(total) % 3 == 0)
    with pytest.raises(ValueError):
        from gym_dockauv_amd.parallel import ShardedStepper
        ShardedStepper(4, 8, lambda a, o: None, "cpu", gather_dtype="fp8")
