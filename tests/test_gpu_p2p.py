"""Peer-to-peer gather (include/dockauv.h "Multi-GPU" block, gym_dockauv_amd/parallel.py: P2PGather) on the GPU:
a single rank, and a rehearsal with two and three rank PROCESSES sharing the box's one GPU (IPC handles, push kernel,
stamps and bounded waits are what runs across GPUs; only the fabric is missing)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_single_rank_push_and_timeout():
    import torch
    from gym_dockauv_amd.parallel import P2PGather
    dev = torch.device("cuda", 0)
    n_local, row = 100, 12
    g = P2PGather(n_local, row, 0, world=1, rank=0, lag=0)
    stream = torch.cuda.current_stream().cuda_stream
    for t in range(5):
        rows = torch.randn((n_local, row), device=dev)
        buf = g.push(rows.data_ptr(), stream)
        torch.cuda.synchronize()
        assert torch.equal(buf, rows)
        assert buf.data_ptr() == g.buffer(t % 2).data_ptr()
    assert g.timed_out() == 0
    g.close()
    with pytest.raises(ValueError):
        P2PGather(3, 5, 0, world=1, rank=0)          # 60-byte slices


def test_wait_is_bounded_when_a_peer_never_signals():
    """A rank of a 2-rank world whose peer never shows up: the wait runs out of spins, sets the peer's bit, and every
    later wait returns at once (the C entry points directly; the peer's flag array is stood in for by local memory)."""
    import ctypes as C
    import torch
    from gym_dockauv_amd import _capi
    lib = _capi.load_library()
    flags, other = C.c_void_p(), C.c_void_p()
    assert lib.dockauv_p2p_alloc(0, 256, 1, C.byref(flags), None) == 0
    assert lib.dockauv_p2p_alloc(0, 256, 1, C.byref(other), None) == 0
    slots = (C.c_void_p * 1)(other.value + 0)        # "peer 1's flags[0]"
    stream = torch.cuda.current_stream().cuda_stream
    status = flags.value + 128
    for stamp in (1, 2, 3):
        assert lib.dockauv_p2p_signal_wait(slots, 1, flags.value, 2, 0, stamp, stamp, 2000, status, stream) == 0
    torch.cuda.synchronize()
    from gym_dockauv_amd.parallel import _DevArray
    st = torch.as_tensor(_DevArray(status, (2,), "<i4"), device="cuda:0").cpu().numpy()
    ot = torch.as_tensor(_DevArray(other.value, (1,), "<i4"), device="cuda:0").cpu().numpy()
    assert st[0] == 0b10 and st[1] == 1              # rank 1 was late, first at stamp 1
    assert ot[0] == 3                                # our stamps still went out
    assert lib.dockauv_p2p_free(flags) == 0 and lib.dockauv_p2p_free(other) == 0
    assert lib.dockauv_p2p_signal_wait(slots, 99, flags.value, 2, 0, 1, 1, 10, status, stream) == -1   # DOCKAUV_E_INVALID


@pytest.mark.parametrize("world", [2, 3])
def test_rank_processes_on_one_gpu_gather_bit_exact(world):
    port = 29700 + (os.getpid() % 200) + world
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-m", "tests.p2p_worker", str(r), str(world), str(port), "640", "12"],
                              cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{out[-3000:]}"
        assert "0 bad" in out
        print("".join(l + "\n" for l in out.splitlines() if " us " in l), end="")    # pace lines (pytest -s)


def test_sharded_env_raises_when_a_peer_stops():
    """Two rank processes on the one GPU; rank 1 stops stepping after five steps: rank 0's ShardedTorchDocking3d.step
    must raise DockAUVError within check_every steps of the time-out instead of handing out rows of an earlier step."""
    port = 29950 + (os.getpid() % 40)
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", P2P_WORKER_MODE="stall")
    procs = [subprocess.Popen([sys.executable, "-m", "tests.p2p_worker", str(r), "2", str(port), "640", "400"],
                              cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=180)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{out[-3000:]}"
        assert "0 bad" in out
    assert "step raised as it must" in outs[0]


def test_sharded_env_checks_the_transport_status():
    """One rank: a time-out word set behind the env's back (what a late peer leaves) surfaces as DockAUVError from
    step() at the next check and from close()."""
    import torch
    from gym_dockauv_amd._capi import DockAUVError
    from gym_dockauv_amd.envs.batched import BASE_CONFIG
    from gym_dockauv_amd.envs.torch_env import ShardedTorchDocking3d
    e = ShardedTorchDocking3d(BASE_CONFIG, num_envs=128, transport="p2p", check_every=4)
    assert e.transport == "p2p"
    e.reset()
    a = torch.zeros((128, e.n_u), device=e.device)
    for _ in range(8):
        e.step(a)
    e.stepper.gather._status_view[0] = 0b10
    with pytest.raises(DockAUVError):
        for _ in range(4):
            e.step(a)
    with pytest.raises(DockAUVError):
        e.close()
    e2 = ShardedTorchDocking3d(BASE_CONFIG, num_envs=128)
    assert e2.transport == "rccl"          # the default (BASELINE.json: RCCL all-gather)
    e2.close()


def test_sequence_argument_checks():
    """dockauv_step_gather_sequence refuses what would race: fewer than 4 plans or one row buffer with riding gathers."""
    import ctypes as C
    import torch
    from gym_dockauv_amd import _capi
    from gym_dockauv_amd.envs.batched import BASE_CONFIG, BatchedDocking3d
    from gym_dockauv_amd.parallel import P2PShardedStepper
    dev = torch.device("cuda", 0)
    env = BatchedDocking3d(BASE_CONFIG, num_envs=128, scenario="SimpleDocking3d", device=0, precision="f32",
                           reset_mode="device", device_seed=3, rng="batched")
    env.reset()
    stream = torch.cuda.current_stream().cuda_stream
    st = P2PShardedStepper(128, env.n_observations + 2, lambda a, o: None, dev, world=1, rank=0)
    actions = torch.zeros((128, env.n_u), device=dev)
    ios = (_capi.StepIO * 3)()
    for i in range(3):
        ios[i].actions, ios[i].obs, ios[i].pack_reward_done = actions.data_ptr(), st.rows2[0].data_ptr(), 1
    lib, g = st.gather.lib, st.gather
    assert lib.dockauv_step_gather_sequence(env._handle, ios, 3, g._plans, 4, 0, 1, stream, stream) == -1
    assert b"alternating row buffers" in lib.dockauv_last_error(env._handle)
    for i in range(3):
        ios[i].obs = st.rows2[i & 1].data_ptr()
    assert lib.dockauv_step_gather_sequence(env._handle, ios, 3, g._plans, 2, 0, 1, stream, stream) == -1
    assert b"at least 4 plans" in lib.dockauv_last_error(env._handle)
    assert lib.dockauv_step_gather_sequence(env._handle, ios, 3, g._plans, 4, 0, 2, stream, stream) == -1
    assert lib.dockauv_step_gather_sequence(env._handle, ios, 3, g._plans, 4, 0, 1, stream, stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(st.bufs[2], st.rows2[0]) and torch.equal(st.bufs[1], st.rows2[1])
    assert g.timed_out() == 0
    st.close()
    env.close()


def test_riding_gather_of_sensor_free_handles_beyond_65536_envs():
    """sensor-free handles beyond 65 536 envs run the 128- / 64-thread kernels, whose default instantiation has
    write-back stores and carries no copy groups; a lag-1 sequence must get the write-through one (every step of the
    sequence launches, the riding gathers deliver the rows bit for bit) -- and bfloat16 rows are refused up front."""
    import torch
    from gym_dockauv_amd import _capi
    from gym_dockauv_amd.envs.batched import BASE_CONFIG, BatchedDocking3d
    from gym_dockauv_amd.parallel import P2PShardedStepper
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    for n in (65536 + 64, 131072 + 64):          # 128 threads per group, 64 threads per group
        env = BatchedDocking3d(BASE_CONFIG, num_envs=n, scenario="SimpleDocking3d", device=0, precision="f32",
                               reset_mode="device", device_seed=3, rng="batched")
        env.reset()
        st = P2PShardedStepper(n, env.n_observations + 2, lambda a, o: None, dev, world=1, rank=0)
        actions = torch.rand((n, env.n_u), device=dev) * 2 - 1
        ios = (_capi.StepIO * 4)()
        for i in range(4):
            ios[i].actions, ios[i].obs, ios[i].pack_reward_done = actions.data_ptr(), st.rows2[i & 1].data_ptr(), 1
        lib, g = st.gather.lib, st.gather
        rc = lib.dockauv_step_gather_sequence(env._handle, ios, 4, g._plans, 4, 0, 1, stream, stream)
        assert rc == 0, lib.dockauv_last_error(env._handle)
        torch.cuda.synchronize()
        assert torch.equal(st.bufs[3], st.rows2[1]) and torch.equal(st.bufs[2], st.rows2[0])
        assert bool(torch.isfinite(st.rows2[1]).all()) and float(st.rows2[1].abs().max()) > 0 and g.timed_out() == 0
        for i in range(4):
            ios[i].pack_reward_done = 2
        assert lib.dockauv_step_gather_sequence(env._handle, ios, 4, g._plans, 4, 4, 1, stream, stream) == -1
        assert b"float32 packed rows" in lib.dockauv_last_error(env._handle)
        st.close()
        env.close()


def test_sharded_env_single_rank_equals_torch_env():
    """ShardedTorchDocking3d with one rank (both transports) hands out what TorchDocking3d does, bit for bit."""
    import numpy as np
    import torch
    from gym_dockauv_amd.envs.batched import BASE_CONFIG
    from gym_dockauv_amd.envs.torch_env import ShardedTorchDocking3d, TorchDocking3d
    dev = torch.device("cuda", 0)
    n = 320
    ref = TorchDocking3d(BASE_CONFIG, num_envs=n, scenario="CapsuleCurrentDocking3d", device_seed=5)
    ref.batch._gen = np.random.default_rng(21)
    ref.reset()
    envs = [ShardedTorchDocking3d(BASE_CONFIG, num_envs=n, scenario="CapsuleCurrentDocking3d", transport=tr,
                                  device_seed=5, host_seed=21) for tr in ("p2p", "rccl")]
    for e in envs:
        assert e.n_local == n and e.first == 0
        assert torch.count_nonzero(e.reset()) == 0
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    for t in range(25):
        a = torch.rand((n, ref.n_u), device=dev, generator=gen) * 2 - 1
        o0, r0, d0 = ref.step(a)
        for e in envs:
            o, r, d = e.step(a)
            assert torch.equal(o.view(torch.int32), o0.contiguous().view(torch.int32)) and torch.equal(r, r0) and torch.equal(d, d0)
    with pytest.raises(ValueError):
        envs[0].step(torch.zeros((n + 1, ref.n_u), device=dev))
    for e in envs:
        e.close()
    ref.close()
