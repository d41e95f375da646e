"""
Full-size parity (What's missing 4 and 5): each BASELINE config at its stated batch size, every env of the
batch against the reference.  The golden steps of the config's trajectory (reference outputs, tests/golden) are tiled
over the batch -- env j starts where golden step j % T started and gets that step's action -- one kernel launch of
the float32 product path steps all of them, and EVERY env must reproduce its golden row (same tolerances and
exclusion rules as the teacher-forced tests of tests/test_gpu_parity.py: 1e-5 on observations).

  config 2: BlueROV2 SimpleDocking3d,            4 096 envs   <- traj_config1_simple_bluerov2
  config 3: BlueROV2 + 16-beam fan + 8 spheres, 65 536 envs   <- traj_SphereDocking3d_bluerov2_fan16(_random)
  config 4: LAUV ObstaclesDocking3d, h = 0.02,  32 768 envs   <- traj_ObstaclesDocking3d_lauv_goto, ..._lauv_near, ..._lauv_ram
  config 5: BlueROV2 / LAUV interleaved 50/50, ObstaclesCurrentDocking3d, h = 0.02, 65 536 envs (VK_MIXED kernel)
            <- traj_ObstaclesCurrentDocking3d_bluerov2_h002_random (even envs) + ..._lauv_random (odd envs), and the
               ..._near pair (vehicles next to an obstacle: 30-50 % of the rays in range)
`done` is compared exactly (helpers.justified_done_flips: enumerated threshold cases only), for both kernels.
"""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


def run_tiled(names, n_envs, precision="f32", threads=0, split=None):
    """names: one golden trajectory, or two for a mixed batch (env j uses names[j % 2]; with `split`: a KIND-SORTED batch, the
    envs [0, split) use names[0], the envs [split, n_envs) names[1])."""
    from gym_dockauv_amd.envs.batched import BatchedDocking3d
    gs = [H.load(n) for n in names]
    K = len(gs)
    cfg = H.config_from_meta(gs[0])
    for g in gs[1:]:
        c2 = H.config_from_meta(g)
        assert c2["t_step_size"] == cfg["t_step_size"] and c2["max_timesteps"] == cfg["max_timesteps"]
        assert c2["radar"] == cfg["radar"] and H.scenario_of(g) == H.scenario_of(gs[0])
    max_caps = max(int(g["ep_n_capsules"].max()) if g["ep_n_capsules"].size else 0 for g in gs)
    max_sph = max(int(g["ep_sph_radii"].shape[1]) for g in gs)
    vehicles = None
    if K == 2:
        kind_of = (lambda j: j % 2) if split is None else (lambda j: int(j >= split))
        vehicles = [str(gs[kind_of(j)]["meta_vehicle"]) for j in range(n_envs)]
        assert sorted(set(vehicles)) == ["BlueROV2", "LAUV"]
    env = BatchedDocking3d(cfg, num_envs=n_envs, scenario=H.scenario_of(gs[0]), precision=precision, auto_reset=False,
                           max_capsules=max_caps, max_spheres=max_sph, current_mu=float(gs[0]["ep_current"][0, 0]),
                           vehicles=vehicles, threads_per_group=threads)
    try:
        # env j -> trajectory j % K, golden step (j // K) % T
        parts = []
        for k, g in enumerate(gs):
            if split is None:
                envs_k = np.arange(k, n_envs, K)
                steps_k = envs_k // K
            else:
                envs_k = np.arange(0, split) if k == 0 else np.arange(split, n_envs)
                steps_k = envs_k - envs_k[0]
            parts.append((envs_k, H.teacher_forced_inputs(g, steps_k % int(g["meta_T"]), max_caps, max_sph)))

        def merge(get):
            first = get(parts[0][1])
            out = np.zeros((n_envs,) + first.shape[1:], dtype=first.dtype)
            for envs_k, inp in parts:
                out[envs_k] = get(inp)
            return out
        inp = {k: merge(lambda p, k=k: p[k]) for k in ("state", "u", "tsteps", "noise", "actions")}
        inp["episodes"] = {k: merge(lambda p, k=k: p["episodes"][k]) for k in parts[0][1]["episodes"]}
        gold = {k: merge(lambda p, k=k: p["gold"][k]) for k in parts[0][1]["gold"]}
        H.load_teacher_forced(env, inp)
        obs, rew, done, _ = env.step(inp["actions"][:, :env.n_u], noise=inp["noise"], extras=True)
        res = H.check_teacher_forced(env, obs, rew, done, gold, precision, "+".join(names) + f" x{n_envs}")
        # ... and the PRODUCT instantiation of the kernel (device pointers, mandatory outputs only: what bench.py and
        # the torch envs launch) from the same inputs: every env against its golden row as well
        ray_outlier = (np.abs(env.intersec_dist - gold["ray_dist"]) > H.TOL[precision]["ray"]).any(axis=1)
        H.load_teacher_forced(env, inp)
        o2, r2, d2 = H.DeviceStepper(env).step(inp["actions"][:, :env.n_u])
        tol = H.TOL[precision]
        # (the packed row carries the reward as float32: its own rounding of up to 2e-7 relative on top)
        H.compare_obs_reward(o2, r2, None, gold, ray_outlier, precision, "+".join(names) + f" x{n_envs} (product kernel)",
                             rew_floor=2e-7)
        # done of the product kernel (it emits no condition bits): exact but for enumerated threshold cases
        H.justified_done_flips(env, d2, gold, precision, "+".join(names) + f" x{n_envs} (product kernel)")
        new_state = env.state
        lin = [0, 1, 2, 6, 7, 8, 9, 10, 11]
        np.testing.assert_allclose(new_state[:, lin], gold["state"][:, lin], rtol=0, atol=tol["state"])
        assert H.angle_diff(new_state[:, 3:6], gold["state"][:, 3:6]).max() <= tol["state"]
        assert np.array_equal(env.t_steps, gold["t_steps"])
        # the batch is made of copies: copies of the same golden step must also agree with each other bit for bit
        T0 = int(gs[0]["meta_T"]) * K
        if n_envs >= 2 * T0 and split is None:
            assert np.array_equal(obs[:T0], obs[T0:2 * T0]) and np.array_equal(rew[:T0], rew[T0:2 * T0])
        return res
    finally:
        env.close()


def test_config2_full_size():
    run_tiled(["traj_config1_simple_bluerov2"], 4096)


@pytest.mark.parametrize("name", ["traj_SphereDocking3d_bluerov2_fan16", "traj_SphereDocking3d_bluerov2_fan16_random"])
def test_config3_full_size(name):
    run_tiled([name], 65536)


def test_config3_write_back_twin_beyond_262144_envs():
    """The 256-thread ray kernel has a twin with write-back stores that launch_vk (dockauv_step.hip.inc) only picks for
    batches beyond 262 144 envs -- launches that run in several rounds of groups (heavy fans, or a caller's choice of 256
    threads: light fans get one wave per group there).  Same source, other stores: here it meets
    the reference's rows like every other product kernel (262 208 envs = 4 097 groups, the smallest batch that selects it)."""
    run_tiled(["traj_SphereDocking3d_bluerov2_fan16_random"], 262144 + 64, threads=256)   # (0 = auto: one wave per group there)


@pytest.mark.parametrize("which,precision", [("sphere", "f32"), ("sphere", "f64"), ("lauv_near", "f32"), ("lauv_ram", "f32"),
                                             ("mixed_near", "f32"), ("mixed_near", "f64")])
def test_one_wave_ray_groups_vs_reference(which, precision):
    """Ray groups of ONE wave (threads_per_group = 64: what dockauv_create picks for light fans beyond 524 288 envs) keep
    pose, slot counts, masks and the collision flag in registers and overlay the observation tile on the obstacle records
    (dockauv_step.hip.inc: SOLO) -- a layout of their own, held to the reference's rows here: 16-lane and 64-lane fans,
    spheres and capsules (with collisions), the mixed kernel, both precisions; batch not a multiple of 64."""
    names = {"sphere": ["traj_SphereDocking3d_bluerov2_fan16_random"], "lauv_near": ["traj_ObstaclesDocking3d_lauv_near"],
             "lauv_ram": ["traj_ObstaclesDocking3d_lauv_ram"], "mixed_near": MIXED_PAIRS["near"]}[which]
    run_tiled(names, 1000, precision, threads=64)


def test_group_shape_the_library_picks():
    """threads_per_group = 0: dockauv_create's choice by workload and batch size (dockauv_capi.hip; the tables behind it:
    profiles/r4/threads_large.txt), read back through dockauv_threads_per_group."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from gym_dockauv_amd.envs.batched import BatchedDocking3d
    expect = {(2, 4096): 256, (2, 131072): 128, (2, 262144): 64,
              (3, 65536): 256, (3, 262144): 64,
              (4, 32768): 512, (4, 65536): 256, (4, 196608): 256, (4, 262144): 64,
              (5, 65536): 256, (5, 262144): 256, (5, 524288): 64}
    for (cid, n), threads in expect.items():
        wl = bench.workload(cid, n)
        env = BatchedDocking3d(wl["cfg"], num_envs=n, scenario=wl["scenario"], precision="f32", reset_mode="device", device_seed=1,
                               rng="batched", vehicles=wl["vehicles"])
        try:
            assert env.threads_in_use == threads, (cid, n, env.threads_in_use, threads)
        finally:
            env.close()
    # float64 has no register-resident records: the one-wave shape of a heavy fan starts where it did before
    wl = bench.workload(4, 262144)
    env = BatchedDocking3d(wl["cfg"], num_envs=262144, scenario=wl["scenario"], precision="f64", reset_mode="device", device_seed=1,
                           rng="batched")
    try:
        assert env.threads_in_use == 256
    finally:
        env.close()


@pytest.mark.parametrize("which", ["lauv_near", "mixed_near"])
def test_one_wave_capsule_groups_at_the_size_that_selects_them(which):
    """Config 4 beyond 196 608 envs and mixed batches beyond 393 216 run ONE wave per group with the completed capsule
    records in registers (dockauv_step.hip.inc: regrec); here at the smallest batches that select that shape by themselves
    (threads_per_group = 0), every env against the reference's rows (ray-rich fixtures)."""
    names, n = {"lauv_near": (["traj_ObstaclesDocking3d_lauv_near"], 196608 + 64),
                "mixed_near": (MIXED_PAIRS["near"], 393216 + 64)}[which]
    run_tiled(names, n)


# "_near": vehicles that start 4-6 m from a capsule and face it -- 30-50 % of all rays in range, every step with hits
# (oracle/gen_golden.py: gen_near_obstacles); the older LAUV trajectories never have a ray in range (min_ray = 10.0)
MIXED_PAIRS = {
    "random": ["traj_ObstaclesCurrentDocking3d_bluerov2_h002_random", "traj_ObstaclesCurrentDocking3d_lauv_random"],
    "near": ["traj_ObstaclesCurrentDocking3d_bluerov2_h002_near", "traj_ObstaclesCurrentDocking3d_lauv_near"],
}


@pytest.mark.parametrize("name", ["traj_ObstaclesDocking3d_lauv_goto", "traj_ObstaclesDocking3d_lauv_near",
                                  "traj_ObstaclesDocking3d_lauv_ram"])
def test_config4_full_size(name):
    g = H.load(name)
    if not name.endswith("_goto"):   # the point of these fixtures: the LAUV 63-ray eight-wave kernel with rays that hit
        assert float((g["ray_dist"] < float(g["meta_radar_max_dist"])).mean()) >= 0.15
    if name.endswith("_ram"):        # ... and with collisions (full thrust into a capsule)
        assert int(g["conditions"][:, 4].sum()) >= 3
    run_tiled([name], 32768)


@pytest.mark.parametrize("pair", sorted(MIXED_PAIRS))
def test_config5_mixed_full_size(pair):
    run_tiled(MIXED_PAIRS[pair], 65536)


@pytest.mark.parametrize("pair,n,split", [("near", 65536, 32768), ("random", 65536, 32768), ("near", 1000, 437), ("near", 262144 + 128, 131072 + 1)])
def test_config5_kind_sorted_vs_reference(pair, n, split):
    """A KIND-SORTED mixed batch (every BlueROV2 env in front of every LAUV env: BatchedDocking3d(sort_vehicles=True)'s device
    layout): every group but one holds ONE vehicle kind, one of the mixed kernel's two integrating waves owns all 64 lanes and
    the other none.  Every env against the reference's rows: config 5's size with the split on a group boundary, a small batch
    with an odd split (a group of both kinds in the middle), and the write-back twin beyond 262 144 envs."""
    run_tiled(MIXED_PAIRS[pair], n, split=split)


@pytest.mark.parametrize("pair", sorted(MIXED_PAIRS))
@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_config5_mixed_batch_vs_reference(precision, pair):
    """The mixed-vehicle kernel (VK_MIXED) directly against the reference's outputs, both precisions, small batch
    (one copy of each golden step; float64 to 1e-9)."""
    run_tiled(MIXED_PAIRS[pair], 480, precision)
