"""GPU: in-kernel episode reset.
  * DOCKAUV_RESET_DEVICE: the Philox4x32-10 draws are bit-exact integer work -> the generated episodes must equal the
    host generators (gym_dockauv_amd/scenarios.py, pinned against the reference's reset draws) fed with the NumPy
    Philox restatement (oracle/philox_ref.py, pinned against the Random123 vectors), to float rounding.
  * DOCKAUV_RESET_POOL: VecEnv semantics -- state comes from the staged pool, returned obs is the reference's reset
    observation (zeros, Q8), the terminal observation is delivered separately."""
import copy

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu

SCN = ["SimpleDocking3d", "SimpleCurrentDocking3d", "CapsuleDocking3d", "CapsuleCurrentDocking3d",
       "ObstaclesDocking3d", "ObstaclesNoCapDocking3d", "ObstaclesCurrentDocking3d"]


def cfg_short(max_t):
    from gym_dockauv_amd.config.env_config import BASE_CONFIG
    cfg = copy.deepcopy(BASE_CONFIG)
    cfg["max_timesteps"] = max_t
    return cfg


@pytest.mark.parametrize("precision", ["f32", "f64"])
@pytest.mark.parametrize("scenario", SCN)
def test_device_reset_matches_philox_reference(scenario, precision):
    from gym_dockauv_amd import _capi, scenarios
    from gym_dockauv_amd.envs.batched import BatchedDocking3d
    from oracle import philox_ref
    N, seed = 333, 0x1234ABCD5678
    env = BatchedDocking3d(cfg_short(2), num_envs=N, scenario=scenario, precision=precision, reset_mode="device",
                           device_seed=seed, rng="batched")
    try:
        env.reset()
        a = np.zeros((N, env.n_u))
        for t in range(3):
            obs, rew, done, infos = env.step(a)
            assert done.all() == (t == 2), f"step {t}: done = {done.sum()}"
        # all envs hit max_timesteps at the third step and were regenerated in-kernel as episode 2
        assert np.all(obs == 0)
        assert "terminal_observation" in infos[0] and np.abs(infos[0]["terminal_observation"]).max() > 0
        U = philox_ref.episode_uniforms(seed, np.arange(N), np.full(N, 2))
        ref = scenarios.episodes_from_uniforms(scenario, U, env.config["max_attitude"], env.config["max_dist_from_goal"],
                                               env.max_capsules, env.max_spheres)
        tol = 1e-12 if precision == "f64" else 2e-6
        st = env.state
        np.testing.assert_allclose(st[:, 0:6], ref["pose"], rtol=0, atol=tol * 15)
        assert np.all(st[:, 6:] == 0) and np.all(env.u == 0)
        goal = env.get_field(_capi.F_GOAL)
        np.testing.assert_allclose(goal[:, 0:3], ref["goal"][:, 0:3], rtol=0, atol=tol * 4)
        d = np.abs(goal[:, 3] - ref["goal"][:, 3])
        assert np.minimum(d, 2 * np.pi - d).max() <= tol * 4
        np.testing.assert_allclose(env.get_field(_capi.F_CURRENT), ref["current"], rtol=0, atol=tol * 4)
        if env.max_capsules:
            np.testing.assert_allclose(env.get_field(_capi.F_CAPSULES), ref["capsules"], rtol=0, atol=tol * 20)
        assert np.all(env.t_steps == 0) and np.all(env.cumulative_reward == 0)
        assert np.all(env.get_field(_capi.F_EPISODE)[:, 0] == 2)
        # the regenerated episodes must be steppable
        obs, rew, done, _ = env.step(a)
        assert np.isfinite(obs).all() and np.isfinite(rew).all() and not done.any()
    finally:
        env.close()


def test_pool_autoreset_vecenv_semantics():
    from gym_dockauv_amd import _capi
    from gym_dockauv_amd.envs.batched import BatchedDocking3d
    N = 130
    env = BatchedDocking3d(cfg_short(1), num_envs=N, scenario="CapsuleCurrentDocking3d", precision="f32",
                           reset_mode="pool", rng="batched")
    try:
        env.reset()
        pool_pose = env.get_field(_capi.F_POOL_POSE)
        pool_goal = env.get_field(_capi.F_POOL_GOAL)
        pool_cur = env.get_field(_capi.F_POOL_CURRENT)
        a = np.random.RandomState(0).uniform(-1, 1, (N, env.n_u))
        obs1, _, done1, _ = env.step(a)
        assert not done1.any() and np.abs(obs1).max() > 0
        # second step: t_steps (1) >= max_timesteps (1) -> done for all, reset from the pool inside the kernel
        env.rng_mode = "frozen"     # keep the host from restaging, so the live state must equal the old pool
        obs2, rew2, done2, infos = env.step(a)
        assert done2.all() and np.all(obs2 == 0)
        term = np.stack([i["terminal_observation"] for i in infos])
        assert np.abs(term).max() > 0 and np.isfinite(term).all()
        assert all(i["conditions_true"] == [3] for i in infos)
        st = env.state
        np.testing.assert_allclose(st[:, 0:6], pool_pose, atol=1e-6)
        assert np.all(st[:, 6:] == 0) and np.all(env.u == 0) and np.all(env.t_steps == 0)
        np.testing.assert_allclose(env.get_field(_capi.F_GOAL), pool_goal, atol=1e-6)
        np.testing.assert_allclose(env.get_field(_capi.F_CURRENT), pool_cur, atol=1e-6)
        assert np.all(env.cumulative_reward == 0)
    finally:
        env.close()


def test_tail_group_and_large_batch():
    """N not a multiple of 64 (partial last workgroup) and a batch large enough to fill the chip several times:
    envs are independent, so a big batch made of copies of a small one must reproduce the small one exactly."""
    from gym_dockauv_amd import _capi
    from gym_dockauv_amd.envs.batched import BatchedDocking3d
    small = BatchedDocking3d(num_envs=77, scenario="ObstaclesDocking3d", reset_mode="none", rng="batched")
    big = BatchedDocking3d(num_envs=77 * 400 + 13, scenario="ObstaclesDocking3d", reset_mode="none", rng="batched")
    try:
        small.reset()
        reps = -(-big.num_envs // 77)
        for f in (_capi.F_STATE, _capi.F_GOAL, _capi.F_CURRENT, _capi.F_CAPSULES):
            big.set_field(f, np.tile(small.get_field(f), (reps, 1))[: big.num_envs])
        rs = np.random.RandomState(3)
        for t in range(5):
            a = rs.uniform(-1, 1, (77, 6))
            o_s, r_s, d_s, _ = small.step(a)
            o_b, r_b, d_b, _ = big.step(np.tile(a, (reps, 1))[: big.num_envs])
            assert np.array_equal(np.tile(o_s, (reps, 1))[: big.num_envs], o_b)
            assert np.array_equal(np.tile(r_s, reps)[: big.num_envs], r_b)
            assert np.array_equal(np.tile(d_s, reps)[: big.num_envs], d_b)
    finally:
        small.close()
        big.close()


SEQ_CASES = {
    # name: (bench config id or scenario kwargs, envs, threads)
    "config2_256": (2, 777, 0), "config2_128": (2, 1000, 128), "config2_64": (2, 1000, 64),
    "config3_256": (3, 777, 0), "config3_64": (3, 1000, 64),
    "config4_512": (4, 777, 512), "config4_256": (4, 777, 256), "config4_64": (4, 1000, 64),
    "config5_256": (5, 778, 0), "config5_64": (5, 778, 64),
}


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(SEQ_CASES))
def test_step_sequence_equals_single_steps(case):
    """dockauv_step_sequence(n) == n x dockauv_step on the same inputs, bit for bit -- as launches queued back to back
    (option off) AND as the resident fast path (default: every group walks its envs through up to 64 steps per launch, no
    launch boundary in between; include/dockauv.h).  70 steps = two resident launches; episodes are short, so in-kernel
    resets happen inside the sequences; every BASELINE workload's kernel, every group shape; packed float32 and bfloat16
    rows; distinct output rows per step and one buffer for all steps (then the last step's rows must be what is left)."""
    import copy
    import sys
    import os
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from gym_dockauv_amd.envs.batched import BatchedDocking3d
    cid, N, threads = SEQ_CASES[case]
    K = 70
    dev = torch.device("cuda", 0)
    wl = bench.workload(cid, N)
    cfg = copy.deepcopy(wl["cfg"])
    cfg["max_timesteps"] = 23
    outs = {}
    for mode in ("single", "launches", "resident", "resident_one_buffer", "resident_bf16", "single_bf16"):
        env = BatchedDocking3d(cfg, num_envs=N, scenario=wl["scenario"], precision="f32", reset_mode="device", device_seed=99,
                               rng="batched", vehicles=wl["vehicles"], threads_per_group=threads)
        env._gen = np.random.default_rng(3)
        env.reset()
        g = torch.Generator(device=dev)
        g.manual_seed(5)
        acts = torch.rand((K, N, env.n_u), device=dev, generator=g) * 2 - 1
        packed = "bf16" if mode.endswith("bf16") else True
        words = env.packed_row_words(packed)
        out = torch.zeros((K, N, words), device=dev, dtype=torch.float32)
        stream = torch.cuda.current_stream().cuda_stream
        if mode.startswith("single"):
            for k in range(K):
                env.step_device(acts[k].data_ptr(), out[k].data_ptr(), stream=stream, packed=packed)
        else:
            env.set_sequence_resident(mode.startswith("resident"))
            one = mode == "resident_one_buffer"
            ios = env.make_step_sequence([acts[k].data_ptr() for k in range(K)],
                                         [out[0 if one else k].data_ptr() for k in range(K)], packed=packed)
            env.run_step_sequence(ios, stream=stream)
        torch.cuda.synchronize()
        env.synchronize()
        outs[mode] = (out.cpu().numpy().view(np.uint32), env.state.copy(), env.get_field(9).copy(), env.t_steps.copy())
        env.close()
    ref = outs["single"]
    assert int(ref[2].max()) >= 2, "episodes must end inside the sequence"       # (episode counters: resets happened)
    for mode in ("launches", "resident"):
        for a, b in zip(ref, outs[mode]):
            assert np.array_equal(a, b), f"{case}: {mode} differs from single launches"
    one = outs["resident_one_buffer"]
    assert np.array_equal(one[0][0], ref[0][K - 1]) and all(np.array_equal(a, b) for a, b in zip(ref[1:], one[1:]))
    for a, b in zip(outs["single_bf16"], outs["resident_bf16"]):
        assert np.array_equal(a, b), f"{case}: resident bf16 rows differ from single launches"


@pytest.mark.gpu
def test_torch_env_matches_host_path():
    """TorchDocking3d (device pointers, packed rows) == BatchedDocking3d.step (host pointers) on the same seeds."""
    import torch
    from gym_dockauv_amd.envs.batched import BatchedDocking3d
    from gym_dockauv_amd.envs.torch_env import TorchDocking3d
    N, K = 300, 40
    tenv = TorchDocking3d(num_envs=N, scenario="CapsuleCurrentDocking3d", device_seed=11)
    henv = BatchedDocking3d(num_envs=N, scenario="CapsuleCurrentDocking3d", precision="f32", reset_mode="device",
                            device_seed=11, rng="batched")
    try:
        tenv.batch._gen = np.random.default_rng(8)
        henv._gen = np.random.default_rng(8)
        o = tenv.reset()
        henv.reset()
        assert o.shape == (N, tenv.n_obs) and not bool(o.any())
        g = torch.Generator(device=tenv.device)
        g.manual_seed(1)
        n_done = 0
        for k in range(K):
            a = torch.rand((N, tenv.n_u), device=tenv.device, generator=g) * 2 - 1
            # even steps ask for terminal observations (product kernel with the terminal copy, TERM), odd steps do not (plain
            # product kernel); the host path runs the full instantiation: separate instantiations of one source contract a
            # few multiply-adds differently, hence 5e-6 instead of bit equality over the free-running steps
            want = k % 2 == 0
            obs, rew, done = tenv.step(a, want_terminal_obs=want)
            ho, hr, hd, infos = henv.step(a.cpu().numpy())
            np.testing.assert_allclose(obs.cpu().numpy(), ho, rtol=0, atol=5e-6)
            np.testing.assert_allclose(rew.cpu().numpy(), hr, rtol=1e-5, atol=1e-5)
            assert np.array_equal(done.cpu().numpy(), hd)
            assert not bool(obs[done].any()), "rows of finished envs must hold the reset observation (zeros)"
            if want:
                for i in np.flatnonzero(hd):
                    term = tenv.terminal_observation[i].cpu().numpy()
                    np.testing.assert_allclose(term, infos[i]["terminal_observation"], rtol=0, atol=5e-6)
                    assert np.abs(term).max() > 0
                n_done += int(hd.sum())
        assert n_done > 0, "the run must cover auto-resets"
    finally:
        tenv.close()
        henv.close()


def _layout_cases():
    import copy
    from gym_dockauv_amd.config.env_config import BASE_CONFIG
    lauv = copy.deepcopy(BASE_CONFIG)
    lauv["vehicle"], lauv["t_step_size"] = "LAUV", 0.02
    h002 = copy.deepcopy(BASE_CONFIG)
    h002["t_step_size"] = 0.02
    fan16 = copy.deepcopy(BASE_CONFIG)
    fan16["radar"].update(alpha=30 * np.pi / 180, beta=30 * np.pi / 180, ray_per_deg=10 * np.pi / 180)
    mixed = ["BlueROV2" if i % 2 == 0 else "LAUV" for i in range(333)]
    # vehicle-sorted: whole 64-env groups of ONE kind inside a mixed batch (an integrating wave of such a group owns no lane)
    sorted_mixed = ["BlueROV2"] * 192 + ["LAUV"] * 141
    base = copy.deepcopy(BASE_CONFIG)
    for c in (base, lauv, h002, fan16):
        c["max_timesteps"] = 17          # every env runs into t_max twice in 40 steps: in-kernel resets in every case
    return [("SimpleCurrentDocking3d", base, None, (64, 128, 256)),
            ("ObstaclesCurrentDocking3d", base, None, (64, 256, 512)),
            ("ObstaclesDocking3d", lauv, None, (64, 256, 512)),
            ("SphereDocking3d", fan16, None, (64, 256, 512)),
            ("ObstaclesCurrentDocking3d", h002, mixed, (64, 256, 512)),
            ("ObstaclesCurrentDocking3d", h002, sorted_mixed, (256,))]


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(6))
def test_wave_layouts_and_product_kernels_agree(case):
    """The product instantiations of the step kernel (device pointers, mandatory outputs only) in every group layout
    -- one wave per group (everything in wave 0) up to eight (bookkeeper / resetter / observation waves, prefetch waves,
    ray passes spread over all, two integrating waves for mixed batches) -- against the full-output instantiation that
    the host-pointer path runs (and that the golden-vector tests check): the same arithmetic on the same inputs, so
    the same trajectories incl. in-kernel resets.  (Not bit for bit: separate instantiations, the compiler contracts a
    few multiply-adds differently, 1-2 ulp per step; the dynamics are damped, so the difference stays at that level.)"""
    from gym_dockauv_amd import _capi
    from gym_dockauv_amd.envs.batched import BatchedDocking3d
    scenario, cfg, vehicles, layouts = _layout_cases()[case]
    N, K = 333, 40

    def make(th):
        env = BatchedDocking3d(cfg, num_envs=N, scenario=scenario, precision="f32", reset_mode="device", device_seed=5,
                               rng="batched", threads_per_group=th, vehicles=vehicles)
        env._gen = np.random.default_rng(4)
        env.reset()
        return env

    rs = np.random.RandomState(9)
    acts = rs.uniform(-1, 1, (K, N, 6))
    ref = make(0)
    try:
        acts = acts[:, :, :ref.n_u]
        tr_ref = [ref.step(acts[k]) for k in range(K)]
        state_ref, goal_ref = ref.state.copy(), ref.get_field(_capi.F_GOAL).copy()
    finally:
        ref.close()
    assert sum(int(t[2].sum()) for t in tr_ref) > 0, "the run must cover in-kernel resets"
    for th in layouts:
        env = make(th)
        try:
            stepper = H.DeviceStepper(env)
            for k in range(K):
                o, r, d = stepper.step(acts[k])
                o1, r1, d1, _ = tr_ref[k]
                assert np.array_equal(d, d1), f"threads {th} step {k}"
                np.testing.assert_allclose(o, o1, rtol=0, atol=2e-6, err_msg=f"threads {th} step {k}")
                np.testing.assert_allclose(r, r1, rtol=2e-6, atol=2e-6)
            np.testing.assert_allclose(env.state, state_ref, rtol=0, atol=5e-6)
            np.testing.assert_allclose(env.get_field(_capi.F_GOAL), goal_ref, rtol=0, atol=5e-6)
        finally:
            env.close()


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_device_noise_matches_philox_reference(precision):
    """dockauv_config.device_noise: the white noise of the Gauss-Markov current (objects/current.py:88) drawn in the
    kernel -- V_c of every env over 30 steps against the host recurrence fed with oracle/philox_ref.py: philox_normal."""
    from gym_dockauv_amd import _capi
    from gym_dockauv_amd.envs.batched import BatchedDocking3d
    from oracle import philox_ref
    N, K, seed = 200, 30, 0xC0FFEE1234
    env = BatchedDocking3d(num_envs=N, scenario="SimpleCurrentDocking3d", precision=precision, reset_mode="none",
                           rng="batched", device_seed=seed, device_noise=True, current_mu=0.01)
    try:
        env._gen = np.random.default_rng(4)
        env.reset()
        rs = np.random.RandomState(1)
        cur = env.get_field(_capi.F_CURRENT)
        cur[:, 0], cur[:, 1], cur[:, 2] = 0.5, 0.2, 1.0          # V_c, V_min, V_max (tests/test_integration.py of the reference)
        env.set_field(_capi.F_CURRENT, cur)
        sigma = rs.uniform(0.0, 0.2, N)
        sigma[::7] = 0.0
        env.set_field(_capi.F_CURRENT_SIGMA, sigma[:, None])
        episode = env.get_field(_capi.F_EPISODE)[:, 0].astype(np.int64)
        h = float(env.config["t_step_size"])
        vc = cur[:, 0].copy()
        for t in range(K):
            env.step(np.zeros((N, 6)))
            z = philox_ref.philox_normal(seed, np.arange(N), episode, np.full(N, t))
            vc = np.clip(vc + (-0.01 * vc + sigma * z) * h, 0.2, 1.0)
            got = env.get_field(_capi.F_CURRENT)[:, 0]
            np.testing.assert_allclose(got, vc, atol=1e-12 if precision == "f64" else 2e-6)
        assert np.ptp(vc[sigma > 0.05]) > 0.01 and np.all(np.abs(vc[::7] - vc[0]) < 1e-9)
    finally:
        env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["traj_ObstaclesCurrentDocking3d_bluerov2_random", "traj_SphereDocking3d_bluerov2_fan16_random",
                                  "traj_ObstaclesDocking3d_lauv_near", "traj_SimpleDocking3d_bluerov2_roll"])
def test_product_kernel_terminal_observation_vs_reference(name):
    """The product kernel with the terminal copy (TERM: packed rows + terminal_obs on device pointers, in-kernel auto-reset)
    against the reference: every env starts where a golden step started; where the reference's episode ends in that step
    the packed row must be the reset observation (zeros, Q8) and terminal_obs the reference's last observation of the
    episode (what SB3's DummyVecEnv hands to train.py:64-71 as infos[i]["terminal_observation"]); everywhere else the
    packed row is the reference's observation and terminal_obs is left untouched."""
    import torch
    from tests import helpers as H
    g = H.load(name)
    T = int(g["meta_T"])
    env, max_caps, max_sph = H.make_batched(g, T, "f32", reset_mode="device", device_seed=3, rng="batched")
    try:
        inp = H.teacher_forced_inputs(g, np.arange(T), max_caps, max_sph)
        H.load_teacher_forced(env, inp)
        dev = torch.device("cuda", env.device)
        n = env.n_observations
        out = torch.zeros((T, n + 2), device=dev)
        term = torch.full((T, n), -7.0, device=dev)
        a = torch.as_tensor(inp["actions"][:, :env.n_u], dtype=torch.float32, device=dev).contiguous()
        env.step_device(a.data_ptr(), out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream, packed=True,
                        terminal_obs_ptr=term.data_ptr())
        torch.cuda.synchronize()
        o, t_ = out.cpu().numpy(), term.cpu().numpy()
        gold = inp["gold"]
        dn = gold["done"].astype(bool)
        assert dn.sum() >= 1, "the trajectory must contain terminal steps"
        assert np.array_equal(o[:, n + 1] > 0.5, dn)
        assert not o[dn, :n].any(), "finished envs: reset observation"
        assert np.all(t_[~dn] == -7.0), "terminal_obs is written only where done"
        d = np.abs(t_[dn] - gold["obs"][dn])
        d[:, 2] = np.minimum(d[:, 2], np.abs(2.0 - d[:, 2]))
        assert d[:, :16].max() <= 1e-5 and d[:, 16:].max(initial=0.0) <= 5e-5, d.max()
        d2 = np.abs(o[~dn, :n] - gold["obs"][~dn])
        d2[:, 2] = np.minimum(d2[:, 2], np.abs(2.0 - d2[:, 2]))
        assert d2[:, :16].max() <= 1e-5
        np.testing.assert_allclose(o[:, n], gold["reward"], rtol=2e-5, atol=2e-5)
    finally:
        env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_vehicle_sorted_batch_equals_caller_ordered_batch(precision, tmp_path):
    """sort_vehicles=True (kind-sorted device ranges for mixed batches) is a property of the product, not
    of the caller: the same mixed batch built with and without it -- same seeds, same actions, in-kernel resets from the
    host-staged pool in the reference's draw order -- hands out identical rows for every env in the CALLER's order (host API:
    step, infos, get_field, reset_envs, episode storage), and the device-pointer path hands out the same rows in device order
    with the permutation to go with them."""
    import pickle
    import torch
    from gym_dockauv_amd import _capi
    from gym_dockauv_amd.config.env_config import BASE_CONFIG
    from gym_dockauv_amd.envs.batched import BatchedDocking3d
    import copy
    cfg = copy.deepcopy(BASE_CONFIG)
    cfg["t_step_size"] = 0.02
    cfg["max_timesteps"] = 17
    N = 203
    rs = np.random.RandomState(11)
    vehicles = [("BlueROV2" if (i % 2 == 0) ^ (rs.rand() < 0.2) else "LAUV") for i in range(N)]
    sel = [3, 64, 65, 130, 202]
    envs = []
    for k, sort in enumerate((False, True)):
        e = BatchedDocking3d(cfg, num_envs=N, scenario="ObstaclesCurrentDocking3d", precision=precision, reset_mode="pool",
                             rng="per_env", vehicles=vehicles, sort_vehicles=sort)
        e.enable_episode_storage(sel, str(tmp_path / f"s{k}"), title="v", capacity=64)
        e.reset(seed=list(range(100, 100 + N)))
        envs.append(e)
    a, b = envs
    try:
        assert not a._sorted and b._sorted
        kinds = np.array([0 if v == "BlueROV2" else 1 for v in vehicles])
        assert np.array_equal(np.sort(b.perm), np.arange(N)) and (np.diff(kinds[b.perm]) >= 0).all()
        assert all(np.diff(b.perm[kinds[b.perm] == k]).min() > 0 for k in (0, 1))          # stable within a kind
        act = np.random.RandomState(5)
        n_done = 0
        for t in range(45):
            x = act.uniform(-1, 1, (N, a.n_u))
            oa, ra, da, ia = a.step(x, extras=True)
            ob, rb, db, ib = b.step(x, extras=True)
            assert np.array_equal(oa, ob) and np.array_equal(ra, rb) and np.array_equal(da, db), f"step {t}"
            assert np.array_equal(a.last_reward_arr, b.last_reward_arr) and np.array_equal(a.intersec_dist, b.intersec_dist)
            for i in np.flatnonzero(da):
                assert ia[i]["conditions_true"] == ib[i]["conditions_true"]
                assert np.array_equal(ia[i]["terminal_observation"], ib[i]["terminal_observation"])
            n_done += int(da.sum())
            if t == 20:   # a host-side reset of a few envs, by the caller's indices
                idx = np.array([0, 1, 64, 150, 202])
                a.reset_envs(idx)
                b.reset_envs(idx)
        assert n_done > N        # every env finished at least once: the pool resets went through the permutation as well
        for f in (_capi.F_STATE, _capi.F_GOAL, _capi.F_CURRENT, _capi.F_CAPSULES, _capi.F_TSTEPS, _capi.F_VEHICLE_ID):
            assert np.array_equal(a.get_field(f), b.get_field(f)), f
        assert np.array_equal(a.get_field(_capi.F_STATE, 60, 11), b.get_field(_capi.F_STATE, 60, 11))
        a.episode_storage.flush()
        b.episode_storage.flush()
        fa, fb = sorted(a.episode_storage.files), sorted(b.episode_storage.files)
        assert len(fa) == len(fb) > 0
        key = lambda f: f.split("__ENV_")[1]
        for x, y in zip(sorted(fa, key=key), sorted(fb, key=key)):
            assert key(x) == key(y)
            px, py = pickle.load(open(x, "rb")), pickle.load(open(y, "rb"))
            assert px["env_index"] == py["env_index"] and np.array_equal(px["vehicle"]["states"], py["vehicle"]["states"])
            assert np.array_equal(px["rewards"], py["rewards"])
        # device-pointer path: device rows = the caller's rows permuted
        if precision == "f32":
            dev = torch.device("cuda", 0)
            x = torch.as_tensor(act.uniform(-1, 1, (N, a.n_u)), dtype=torch.float32, device=dev)
            pa = torch.zeros((N, a.n_observations + 2), device=dev)
            pb = torch.zeros_like(pa)
            s = torch.cuda.current_stream().cuda_stream
            perm = torch.as_tensor(b.perm, device=dev)
            a.step_device(x.data_ptr(), pa.data_ptr(), stream=s, packed=True)
            b.step_device(x[perm].contiguous().data_ptr(), pb.data_ptr(), stream=s, packed=True)
            torch.cuda.synchronize()
            assert torch.equal(pa[perm].view(torch.int32), pb.view(torch.int32))
    finally:
        a.close()
        b.close()


@pytest.mark.gpu
def test_torch_envs_hand_back_the_vehicle_permutation():
    """TorchDocking3d / ShardedTorchDocking3d (one rank) with sort_vehicles=True: rows come out in device order together with
    `perm` (row j = the caller's env perm[j]) and `vehicles_by_row`; permuted back they are the unsorted env's rows, bit for bit."""
    import copy
    import torch
    from gym_dockauv_amd.config.env_config import BASE_CONFIG
    from gym_dockauv_amd.envs.torch_env import ShardedTorchDocking3d, TorchDocking3d
    cfg = copy.deepcopy(BASE_CONFIG)
    cfg["t_step_size"] = 0.02
    N = 300
    kinds = ["BlueROV2" if (i * 7) % 3 else "LAUV" for i in range(N)]
    ref = TorchDocking3d(cfg, num_envs=N, scenario="ObstaclesCurrentDocking3d", reset_mode="none", vehicles=kinds)
    srt = TorchDocking3d(cfg, num_envs=N, scenario="ObstaclesCurrentDocking3d", reset_mode="none", vehicles=kinds, sort_vehicles=True)
    shd = ShardedTorchDocking3d(cfg, num_envs=N, scenario="ObstaclesCurrentDocking3d", vehicles=kinds, sort_vehicles=True, host_seed=4)
    try:
        for e in (ref, srt):
            e.batch._gen = np.random.default_rng(4)
        ref.reset()
        srt.reset()
        shd.reset()
        # the same episodes everywhere: copy the unsorted env's fields through the caller-ordered host API
        for f in (0, 2, 3, 5):   # state, goal, current, capsules
            rows = ref.batch.get_field(f)
            for e in (ref, srt, shd):   # (ref too: a read-back current has float32 angles, its direction is recomputed from them)
                e.batch.set_field(f, rows)
        perm = srt.perm
        assert srt.vehicles_by_row == [kinds[int(i)] for i in perm.cpu()] and srt.vehicles_by_row == sorted(kinds)
        assert torch.equal(perm.cpu(), torch.as_tensor(shd.perm)) and shd.vehicles_by_row == srt.vehicles_by_row
        g = torch.Generator(device=ref.device)
        g.manual_seed(1)
        for t in range(12):
            a = torch.rand((N, ref.n_u), device=ref.device, generator=g) * 2 - 1
            o0, r0, d0 = ref.step(a)
            o1, r1, d1 = srt.step(a[perm].contiguous())
            o2, r2, d2 = shd.step(a[perm].contiguous())
            assert torch.equal(o0[perm], o1) and torch.equal(r0[perm], r1) and torch.equal(d0[perm], d1), f"TorchDocking3d, step {t}"
            # (the sharded env resets in-kernel, the other two do not: compared while no episode has ended)
            assert not bool(d0.any()), "choose fewer steps: an episode ended"
            assert torch.equal(o0[perm], o2) and torch.equal(r0[perm], r2) and torch.equal(d0[perm], d2), f"ShardedTorchDocking3d, step {t}"
    finally:
        ref.close()
        srt.close()
        shd.close()
