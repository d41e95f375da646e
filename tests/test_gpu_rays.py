"""Ray fan vs capsules / spheres and body collision on the HIP path against the oracle, on RANDOM geometry
(SURVEY.md section 8c, G4/G5): vehicles inside, beside and far from obstacles, obstacles behind the fan, capsules
of every orientation, cap hits, empty slots.  One step from rest with zero action; both sides start from the same
state, so the rays are cast from the same post-step pose (to 1e-7)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 384
MAX_CAP, MAX_SPH = 3, 3


def _scene(rs):
    """Per-env pose + obstacles; every 4th env has the vehicle inside an obstacle, every 5th has empty slots."""
    pos = rs.uniform(-8, 8, (N, 3))
    att = np.stack([rs.uniform(-0.6, 0.6, N), rs.uniform(-0.6, 0.6, N), rs.uniform(-np.pi, np.pi, N)], axis=1)
    caps = np.zeros((N, MAX_CAP, 7))
    sph = np.zeros((N, MAX_SPH, 4))
    for i in range(N):
        for c in range(MAX_CAP):
            ctr = pos[i] + rs.uniform(-9, 9, 3)
            half = rs.normal(size=3)
            half *= rs.uniform(0.2, 12) / np.linalg.norm(half)
            caps[i, c] = [*(ctr - half), *(ctr + half), rs.uniform(0.5, 3.0)]
        for s in range(MAX_SPH):
            sph[i, s] = [*(pos[i] + rs.uniform(-9, 9, 3)), rs.uniform(0.5, 3.5)]
        if i % 4 == 0:        # vehicle inside capsule 0 / sphere 0
            caps[i, 0, 0:3] = pos[i] + [0.2, -0.1, -3.0]
            caps[i, 0, 3:6] = pos[i] + [0.1, 0.3, 4.0]
            caps[i, 0, 6] = 1.5
            sph[i, 0] = [*(pos[i] + [0.3, 0.2, -0.1]), 2.0]
        if i % 4 == 1:        # capsule straight ahead, axis nearly along the viewing direction (cap hits)
            fwd = np.array([np.cos(att[i, 2]) * np.cos(att[i, 1]), np.sin(att[i, 2]) * np.cos(att[i, 1]), -np.sin(att[i, 1])])
            caps[i, 1, 0:3] = pos[i] + 6 * fwd
            caps[i, 1, 3:6] = pos[i] + 11 * fwd + rs.uniform(-0.3, 0.3, 3)
            caps[i, 1, 6] = 1.2
        if i % 5 == 0:        # trailing empty slots (radius <= 0 closes the list)
            caps[i, 2, 6] = -1.0
            sph[i, 1:, 3] = -1.0
    return pos, att, caps, sph


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_random_obstacle_geometry_vs_oracle(precision):
    from gym_dockauv_amd import _capi
    from gym_dockauv_amd.envs.batched import BatchedDocking3d
    from oracle import dockauv_oracle as orc
    rs = np.random.RandomState(7)
    pos, att, caps, sph = _scene(rs)
    env = BatchedDocking3d(num_envs=N, scenario="CapsuleDocking3d", precision=precision, reset_mode="none",
                           rng="batched", max_capsules=MAX_CAP, max_spheres=MAX_SPH)
    try:
        env.reset()
        state = np.zeros((N, 12))
        state[:, 0:3], state[:, 3:6] = pos, att
        goal = np.concatenate([pos + rs.uniform(-5, 5, (N, 3)), np.zeros((N, 1))], axis=1)
        env.set_field(_capi.F_STATE, state)
        env.set_field(_capi.F_GOAL, goal)
        env.set_field(_capi.F_CURRENT, np.zeros((N, 5)))
        env.set_field(_capi.F_CAPSULES, caps.reshape(N, -1))
        env.set_field(_capi.F_SPHERES, sph.reshape(N, -1))
        obs, rew, done, _ = env.step(np.zeros((N, 6)), extras=True)
        d_gpu = np.asarray(env.intersec_dist, dtype=np.float64)
        col_gpu = env.conditions[:, 4]
        new_state = env.state
    finally:
        env.close()

    d_ref = np.zeros_like(d_gpu)
    col_ref = np.zeros(N, dtype=bool)
    obs_ref = np.zeros_like(obs)
    oracles_max_dist = orc.OracleEnv("CapsuleDocking3d").fan.max_dist
    assert oracles_max_dist == env.radar.max_dist
    for i in range(N):
        o = orc.OracleEnv("CapsuleDocking3d")
        n_c = MAX_CAP if caps[i, 2, 6] > 0 else 2
        n_s = MAX_SPH if sph[i, 1, 3] > 0 else 1
        ep = orc.Episode(position=pos[i], attitude=att[i], goal=goal[i, 0:3], heading_goal=0.0,
                         current=orc.CurrentState(), capsules=[(caps[i, c, 0:3], caps[i, c, 3:6], caps[i, c, 6]) for c in range(n_c)],
                         sphere_centers=sph[i, :n_s, 0:3], sphere_radii=sph[i, :n_s, 3])
        o.reset(episode=ep)
        oo, _, _, _ = o.step(np.zeros(6), noise=0.0)
        d_ref[i], col_ref[i], obs_ref[i] = o.intersec_dist, o.collision, oo
        assert np.abs(o.state[0:3] - new_state[i, 0:3]).max() < (1e-9 if precision == "f64" else 2e-6)

    md = float(oracles_max_dist)                      # radar max_dist: a clamped distance < md is a hit in range
    err = np.abs(d_gpu - d_ref)
    hit_ref, hit_gpu = d_ref < md, d_gpu < md
    assert hit_ref.mean() > 0.15, f"the scene must exercise the intersection code: {hit_ref.mean():.3f} of the rays hit"
    if precision == "f64":
        assert err.max() < 1e-8, f"max ray error {err.max()}"
        assert np.array_equal(col_gpu, col_ref)
        np.testing.assert_allclose(obs, obs_ref, atol=1e-6)
    else:
        # float32, statistics over the rays that HIT (misses are exact by construction): hit/miss flips only at
        # grazing incidence, p99 of the hit distances 5e-5 m, < 0.2 % of the hits off by more than 2e-4 m
        flips = hit_ref != hit_gpu
        assert flips.sum() <= 2e-3 * hit_ref.sum(), f"{flips.sum()} hit/miss flips among {hit_ref.sum()} hits"
        both = hit_ref & hit_gpu
        assert both.sum() > 0.15 * d_ref.size
        assert np.percentile(err[both], 99) < 5e-5, np.percentile(err[both], 99)
        assert (err[both] > 2e-4).mean() < 2e-3, (err[both] > 2e-4).mean()
        assert (col_gpu != col_ref).sum() <= 1
        ok = ~(err > 2e-4).any(axis=1) & (col_gpu == col_ref)
        np.testing.assert_allclose(obs[ok], obs_ref[ok], atol=2e-5)


@pytest.mark.parametrize("blk,ray_per_deg_deg,alpha_deg,beta_deg",
                         [(1, 10, 20, 40), (3, 10, 60, 80), (2, 5, 20, 30), (4, 15, 60, 90),
                          (2, 5, 40, 60),      # 9 x 13 = 117 rays: the lane = env / wave = cell stage of fans wider than 64 rays
                          (2, 10, 10, 30),     # 2 x 4 = 8 rays: 8 envs per wave pass
                          (1, 20, 20, 20)])    # 2 x 2 = 4 rays: 16 envs per wave pass
def test_fan_and_block_sizes_vs_oracle(blk, ray_per_deg_deg, alpha_deg, beta_deg):
    """Ray fans and block-max sizes other than the default 7 x 9 / 2 x 2 (cells with 1, 9 and 16 rays, ragged edge
    cells, 1 to 117 rays: both ray-stage mappings, 1 to 64 envs per wave pass) against the oracle, float64, a few
    free-running steps."""
    import copy
    from gym_dockauv_amd.config.env_config import BASE_CONFIG
    from gym_dockauv_amd.envs.batched import BatchedDocking3d
    from oracle import dockauv_oracle as orc
    cfg = copy.deepcopy(BASE_CONFIG)
    d2r = np.pi / 180
    cfg["radar"].update(alpha=alpha_deg * d2r, beta=beta_deg * d2r, ray_per_deg=ray_per_deg_deg * d2r, blocksize_reduce=blk)
    n = 48
    env = BatchedDocking3d(cfg, num_envs=n, scenario="ObstaclesDocking3d", precision="f64", reset_mode="none", rng="per_env")
    try:
        env.reset(seed=list(range(100, 100 + n)))
        oracles = [orc.OracleEnv("ObstaclesDocking3d", {"radar": dict(cfg["radar"])}) for _ in range(n)]
        for i, o in enumerate(oracles):
            o.reset(seed=100 + i)
        assert env.n_observations == oracles[0].n_obs
        rs = np.random.RandomState(1)
        for t in range(4):
            a = rs.uniform(-1, 1, (n, 6))
            obs, rew, done, _ = env.step(a, extras=True)
            for i, o in enumerate(oracles):
                oo, rr, dd, _ = o.step(a[i])
                np.testing.assert_allclose(env.intersec_dist[i], o.intersec_dist, atol=1e-8)
                np.testing.assert_allclose(obs[i], oo, atol=1e-6)
                assert abs(float(rew[i]) - rr) < 1e-8 * max(1.0, abs(rr)) and bool(done[i]) == dd
    finally:
        env.close()
