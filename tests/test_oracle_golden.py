"""
Pins oracle/dockauv_oracle.py (the NumPy CPU restatement) against
  (a) the golden vectors generated from the reference (tests/golden/*.npz, oracle/gen_golden.py), and
  (b) the known-answer values of the reference's own unit tests.
CPU only.
"""
import glob
import os

import numpy as np
import pytest

from oracle import dockauv_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
VEH = {
    "bluerov2": ("bluerov2", None),
    "bluerov2_direct": ("bluerov2_direct", None),
    "bluerov2_testxml": ("bluerov2", orc.BLUEROV2_TEST_PARAMS),
    "lauv": ("lauv", None),
}


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


# ---------------------------------------------------------------- reference unit-test known answers
def test_kat_geomutils():
    # tests/utils/test_geomutils.py:9-40
    x = orc.ssa(np.array([3 * np.pi, 3 * np.pi - 0.001, np.pi / 2, 0, -4 / 3 * np.pi, 10 / 3 * np.pi]))
    np.testing.assert_allclose(x, [-np.pi, np.pi - 0.001, np.pi / 2, 0, 2 / 3 * np.pi, -2 / 3 * np.pi], atol=1e-7)
    R = orc.rot_zyx(np.pi / 4, np.pi / 4, np.pi / 4)
    np.testing.assert_allclose(R @ [1, 0, 0], [0.5, 0.5, -1 / 2 ** 0.5], atol=1e-7)
    T = orc.t_zyx(np.pi / 4, np.pi / 4)
    np.testing.assert_allclose(T @ [1, 0, 0], [1, 0, 0], atol=1e-7)
    np.testing.assert_allclose(T @ [0, 1, 0], [1 / 2 ** 0.5, 1 / 2 ** 0.5, 1], atol=1e-7)


def test_kat_bluerov2_matrices():
    # tests/objects/test_BlueROV2.py:74-114 (fixture: test_BlueROV2.xml added mass, nu_r = [3,2,1,.3,.2,.1])
    v = orc.VehicleModel("bluerov2", orc.BLUEROV2_TEST_PARAMS)
    nu = np.array([3, 2, 1, 0.3, 0.2, 0.1])
    # C_A from -S(M_A11 nu1) blocks: entries 14.57, 25.4, -0.036
    a1 = v.ma_diag[0:3] * nu[0:3]
    a2 = v.ma_diag[3:6] * nu[3:6]
    C_A = np.zeros((6, 6))
    C_A[0:3, 3:6] = -orc.skew(a1)
    C_A[3:6, 0:3] = -orc.skew(a1)
    C_A[3:6, 3:6] = -orc.skew(a2)
    assert C_A[0, 4] == pytest.approx(14.57)
    assert C_A[2, 3] == pytest.approx(25.4)
    assert C_A[5, 4] == pytest.approx(-0.036)
    assert v.I_b[0, 0] == pytest.approx(0.2146)
    assert v.I_b[1, 1] == pytest.approx(0.2496)
    assert v.I_b[2, 2] == pytest.approx(0.245)
    # C_RB entries 0.023 / -0.069 / -0.06438 via the block formula
    S2, Sg = orc.skew(nu[3:6]), orc.skew(v.r_G)
    C_RB = np.zeros((6, 6))
    C_RB[0:3, 0:3] = v.m * S2
    C_RB[0:3, 3:6] = -v.m * S2 @ Sg
    C_RB[3:6, 0:3] = v.m * Sg @ S2
    C_RB[3:6, 3:6] = -orc.skew(v.I_b @ nu[3:6])
    assert C_RB[0, 3] == pytest.approx(0.023)
    assert C_RB[2, 3] == pytest.approx(-0.069)
    assert C_RB[5, 4] == pytest.approx(-0.06438)
    # the oracle's cross-product form equals (C_RB + C_A) nu
    np.testing.assert_allclose(v.coriolis_force(nu), (C_RB + C_A) @ nu, rtol=1e-13, atol=1e-13)
    g0 = v.restoring(0.0, 0.0)
    assert g0[0] == 0 and g0[1] == 0 and g0[2] != 0
    g1 = v.restoring(0.3, 0.2)
    assert g1[3] != 0 and g1[4] != 0 and g1[5] == 0


def test_kat_unnormalize():
    # tests/objects/test_BlueROV2.py:139-148
    v = orc.VehicleModel("bluerov2", orc.BLUEROV2_TEST_PARAMS)
    v.u_bound = np.array([[-5, 5], [-5, 5], [-5, 5], [-1, 3], [-1, 1], [-1, 1]], dtype=float)
    out = v.unnormalize(np.array([-1.0, -0.5, 0.0, 0.5, 0.5, 1.0]))
    np.testing.assert_array_equal(out, [-5, -2.5, 0.0, 2.0, 0.5, 1.0])


def test_kat_sim_ode_vs_scipy():
    # tests/objects/test_BlueROV2.py:150-188: 100 steps at h=0.01 equal scipy RK45 to 6 decimals
    from scipy.integrate import solve_ivp
    v = orc.VehicleModel("bluerov2", orc.BLUEROV2_TEST_PARAMS)
    v.B_const = np.identity(6)
    v.u_bound = np.array([[-5, 5], [-5, 5], [-5, 5], [-1, 3], [-1, 1], [-1, 1]], dtype=float)
    action = np.array([1, 0, 0, -0.5, 0, 0], dtype=float)
    h = 0.01
    state, u = np.zeros(12), np.zeros(6)
    for _ in range(100):
        state, u, _ = orc.auv_step(v, state, u, action, np.zeros(6), h)
    s2, u2 = np.zeros(12), np.zeros(6)
    a = orc.lowpass_alpha(h)
    for _ in range(100):
        u2 = a * v.unnormalize(action) + (1 - a) * u2
        res = solve_ivp(lambda t, y: v.state_dot(y, u2, np.zeros(6)), [0, h], s2, t_eval=[h], method="RK45")
        s2 = res.y.flatten()
    np.testing.assert_array_almost_equal(s2, state, decimal=6)
    g = load("g3_auv_step")
    np.testing.assert_allclose(state, g["test_sim_ode_final_state"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(u, g["test_sim_ode_final_u"], rtol=1e-13)


def test_kat_shape():
    # tests/objects/test_shape.py:20-103
    point, l11, l12 = np.array([0.5, 0.5, 0.5]), np.array([1., 1, 1]), np.array([1., 1, 0])
    point2, l21, l22 = np.array([-1, -1, -2.5]), np.zeros(3), np.array([2.0, 2.0, 0.0])
    assert orc.seg_point_distance(point, l11, l12) == pytest.approx(0.5 ** 0.5)
    assert orc.seg_point_distance(point2, l21, l22) == pytest.approx(8.25 ** 0.5)
    assert orc.seg_point_distance(point, l11, l12) <= 1 + 0.5
    assert not (orc.seg_point_distance(point2, l21, l22) <= 1 + 0.5)
    assert orc.ray_capsule(l21, l22 - l21, l11, l12, 1.0) == pytest.approx(2 ** 0.5 - 1)
    assert orc.ray_capsule(l21, np.array([-2.0, -2.0, 0.0]), l11, l12, 1.0) == pytest.approx(-(2 ** 0.5 + 1))
    assert orc.ray_capsule(l21, np.array([-2.0, 2.0, 0.0]), l11, l12, 1.0) == -np.inf
    l1 = np.array([[0, 0, 3], [0, -2, 0], [2, 2, 0], [-5, 0, 0]], dtype=float)
    ld = np.array([[0, 0, -2], [0, 1, 0], [1, 0, 0], [1, 0, 0]], dtype=float)
    center = np.array([[0, 0, 0], [-2, 0, 0]], dtype=float)
    rad = np.array([1, 0.5])
    d = [orc.ray_spheres(l1[i], ld[i], center, rad) for i in range(4)]
    assert d[0] == pytest.approx(2.0) and d[1] == pytest.approx(1.0) and d[2] == -np.inf and d[3] == pytest.approx(2.5)
    r = orc.vec_line_point(np.array([0., 0, 1]), np.array([-2., 1, 2]), np.array([2., 1, 0]))
    np.testing.assert_allclose(r, [0, 1, 0], atol=1e-12)


def test_kat_current_ned():
    # tests/objects/test_current.py:25-30
    c = orc.CurrentState(mu=0.01, V_min=0.5, V_max=1.0, V_c=0.5, alpha=np.pi / 4, beta=np.pi / 4, sigma=0.1)
    np.testing.assert_allclose(c.ned(), [0.25, 1 / (2 * 2 ** 0.5), 0.25], atol=1e-12)


def test_kat_sensor_shapes():
    # tests/objects/test_sensor.py:8-24 (shape checks) + SURVEY 3.4 counts
    f = orc.RayFan(30 * np.pi / 180, 20 * np.pi / 180, 5 * np.pi / 180, 5)
    assert f.n_rays == f.alpha.shape[0] == f.beta.shape[0] == f.rd_b.shape[0] and f.rd_b.shape[1] == 3
    f63 = orc.RayFan(60 * np.pi / 180, 80 * np.pi / 180, 10 * np.pi / 180, 10)
    assert (f63.n_v, f63.n_h, f63.n_rays, f63.n_rays_reduced) == (7, 9, 63, 20)
    f16 = orc.RayFan(30 * np.pi / 180, 30 * np.pi / 180, 10 * np.pi / 180, 10)
    assert (f16.n_rays, f16.n_rays_reduced) == (16, 4)


def test_anchor_values_survey():
    # SURVEY.md 8c anchor values
    v = orc.VehicleModel("bluerov2")
    M = v.M_RB + v.M_A
    np.testing.assert_allclose(np.diag(M), [19.07, 19.07, 19.07, 0.3346, 0.3696, 0.365], atol=1e-4)
    assert M[0, 4] == pytest.approx(0.23) and M[1, 3] == pytest.approx(-0.23)
    assert v.M_inv[3, 3] == pytest.approx(3.013628, abs=1e-5)
    env = orc.OracleEnv("SimpleDocking3d")
    env.reset(seed=0)
    np.testing.assert_allclose(env.state[0:3], [12.369163, 5.906876, 6.092013], atol=1e-6)
    np.testing.assert_allclose(env.state[3:6], [-0.111928, 0.213892, -0.392151], atol=1e-6)
    acts = np.random.RandomState(123).uniform(-1, 1, (3, 6))
    for a in acts:
        o, r, d, _ = env.step(a)
    assert r == pytest.approx(-1.644642386865335, abs=1e-12)
    np.testing.assert_allclose(o[0:3], [0.921812, -0.155816, -0.734059], atol=1e-6)
    assert np.all(o[16:] == 1.0)


# ---------------------------------------------------------------- golden vectors from the reference import
@pytest.mark.parametrize("name", list(VEH))
def test_g1_constants(name):
    g = load("g1_constants")
    kind, params = VEH[name]
    v = orc.VehicleModel(kind, params)
    nu0 = g["nu0"]
    np.testing.assert_allclose(v.M_RB, g[name + "_M_RB"], rtol=1e-14, atol=1e-15)
    np.testing.assert_allclose(v.M_A, g[name + "_M_A"], rtol=0, atol=0)
    np.testing.assert_allclose(v.M_inv, g[name + "_M_inv"], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(v.I_b, g[name + "_I_b"], rtol=1e-14, atol=1e-16)
    np.testing.assert_allclose([v.W, v.BY], g[name + "_W_BY"], rtol=1e-15)
    np.testing.assert_allclose(v.input_matrix(nu0), g[name + "_B_at_nu0"], rtol=1e-14)
    np.testing.assert_allclose(v.u_bound, g[name + "_u_bound"], rtol=1e-15)
    assert orc.lowpass_alpha(0.1) == pytest.approx(float(g[name + "_alpha_h0p1"][0]), rel=1e-15)
    np.testing.assert_allclose(v.coriolis_force(nu0), g[name + "_C_at_nu0"] @ nu0, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(v.damping_matrix(nu0), g[name + "_D_at_nu0"], rtol=1e-14, atol=1e-15)
    np.testing.assert_allclose(v.restoring(0.3, -0.2), g[name + "_G_at_eta0"], rtol=1e-13, atol=1e-14)


@pytest.mark.parametrize("name", list(VEH))
def test_g2_state_dot(name):
    g = load("g2_state_dot")
    kind, params = VEH[name]
    v = orc.VehicleModel(kind, params)
    st, us, nc, sd = (g[name + s] for s in ("_state", "_u", "_nu_c", "_state_dot"))
    for i in range(st.shape[0]):
        out = v.state_dot(st[i], us[i], nc[i])
        np.testing.assert_allclose(out, sd[i], rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize("name", list(VEH))
def test_g3_auv_step(name):
    g = load("g3_auv_step")
    kind, params = VEH[name]
    v = orc.VehicleModel(kind, params)
    hs = [0.1, 0.05, 0.01] if name != "lauv" else [0.02, 0.01]
    for h in hs:
        tag = f"{name}_h{h}"
        for i in range(g[tag + "_state"].shape[0]):
            ns, nu, nsd = orc.auv_step(v, g[tag + "_state"][i], g[tag + "_u_prev"][i], g[tag + "_action"][i],
                                       g[tag + "_nu_c"][i], h)
            np.testing.assert_allclose(nu, g[tag + "_new_u"][i], rtol=1e-13, atol=1e-14)
            np.testing.assert_allclose(ns, g[tag + "_new_state"][i], rtol=1e-9, atol=1e-10)
            np.testing.assert_allclose(nsd, g[tag + "_new_state_dot"][i], rtol=1e-8, atol=1e-8)


def test_g4_rays():
    g = load("g4_rays")
    for i in range(g["cap_cap1"].shape[0]):
        for k in range(g["cap_origins"].shape[1]):
            d = orc.ray_capsule(g["cap_origins"][i, k], g["cap_dirs"][i, k], g["cap_cap1"][i], g["cap_cap2"][i],
                                g["cap_rad"][i])
            np.testing.assert_allclose(d, g["cap_dist"][i, k], rtol=1e-10, atol=1e-10)
    for k in range(g["edge_origins"].shape[0]):
        d = orc.ray_capsule(g["edge_origins"][k], g["edge_dirs"][k], g["edge_cap1"], g["edge_cap2"],
                            float(g["edge_rad"][0]))
        np.testing.assert_allclose(d, g["edge_dist"][k], rtol=1e-10, atol=1e-10, equal_nan=True)
    for i in range(g["sph_centers"].shape[0]):
        for k in range(g["sph_origins"].shape[1]):
            d = orc.ray_spheres(g["sph_origins"][i, k], g["sph_dirs"][i, k], g["sph_centers"][i], g["sph_radii"][i])
            np.testing.assert_allclose(d, g["sph_dist"][i, k], rtol=1e-10, atol=1e-10)
    assert np.isfinite(g["cap_dist"]).sum() > 100 and np.isfinite(g["sph_dist"]).sum() > 100
    assert (g["cap_dist"][np.isfinite(g["cap_dist"])] < 0).any()   # "capsule behind" negative distances exist


def test_g4_radar_layout():
    g = load("g4_radar_layout")
    for tag, (al, be, rp) in {"fan63": (60, 80, 10), "fan16": (30, 30, 10), "fan_test": (30, 20, 5)}.items():
        f = orc.RayFan(al * np.pi / 180, be * np.pi / 180, rp * np.pi / 180, 10)
        assert [f.n_v, f.n_h, f.n_rays, f.n_rays_reduced] == g[tag + "_shape"].tolist()
        np.testing.assert_allclose(f.alpha, g[tag + "_alpha"], atol=1e-15)
        np.testing.assert_allclose(f.beta, g[tag + "_beta"], atol=1e-15)
        np.testing.assert_allclose(f.rd_b, g[tag + "_rd_b"], atol=1e-15)
        np.testing.assert_allclose(f.directions_ned(g[tag + "_att"]), g[tag + "_rd_n_att"], atol=1e-14)
        d = f.clamp(g[tag + "_d_in"])
        np.testing.assert_array_equal(d, g[tag + "_d_clamped"])
        np.testing.assert_array_equal(f.reduce(d), g[tag + "_d_reduced"])
        oa = orc.obstacle_avoidance(f.alpha, f.beta, d, f.alpha_max, f.beta_max, f.max_dist)
        assert oa == pytest.approx(float(g[tag + "_oa"][0]), rel=1e-13)


def test_g5_collision():
    g = load("g5_collision")
    for i in range(g["pos"].shape[0]):
        d = orc.seg_point_distance(g["pos"][i], g["cap1"][i], g["cap2"][i])
        assert d == pytest.approx(float(g["seg_dist"][i]), rel=1e-12)
        assert (d <= g["rad"][i] + 1.0) == bool(g["hit_capsule"][i])
        hs = bool(np.any(np.linalg.norm(g["sph_centers"][i] - g["pos"][i][None, :], axis=1) <= 1.0 + g["sph_radii"][i]))
        assert hs == bool(g["hit_spheres"][i])
        np.testing.assert_allclose(orc.vec_line_point(g["pos"][i], g["cap1"][i], g["cap2"][i]),
                                   g["vec_line_point"][i], atol=1e-12)
    assert g["hit_capsule"].any() and (~g["hit_capsule"]).any() and g["hit_spheres"].any()


# ---------------------------------------------------------------- G7 / G8: env-level trajectories
TRAJ = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "traj_*.npz")))


def env_from_meta(g):
    cfg = {
        "vehicle": str(g["meta_vehicle"]),
        "t_step_size": float(g["meta_t_step_size"]),
        "max_timesteps": int(g["meta_max_timesteps"]),
        "reward_set": int(g["meta_reward_set"]),
        "radar": {"alpha": float(g["meta_radar_alpha"]), "beta": float(g["meta_radar_beta"]),
                  "ray_per_deg": float(g["meta_radar_ray_per_deg"]), "max_dist": float(g["meta_radar_max_dist"])},
    }
    name = str(g["meta_env"])
    scenario = name if name in orc.SCENARIOS else "SimpleDocking3d"
    return orc.OracleEnv(scenario, cfg), name


def episode_from_golden(g, e):
    cur = g["ep_current"][e]
    n = int(g["ep_n_capsules"][e])
    caps = [(c[0:3].copy(), c[3:6].copy(), float(c[6])) for c in g["ep_capsules"][e][:n]]
    return orc.Episode(position=g["ep_position"][e], attitude=g["ep_attitude"][e], goal=g["ep_goal"][e],
                       heading_goal=float(g["ep_heading_goal"][e]),
                       current=orc.CurrentState(mu=cur[0], V_min=cur[1], V_max=cur[2], V_c=cur[3], alpha=cur[4],
                                                beta=cur[5], sigma=cur[6]),
                       capsules=caps, sphere_centers=g["ep_sph_centers"][e], sphere_radii=g["ep_sph_radii"][e])


def reset_like_reference(env, g, e, env_name, seed=None):
    """The seven reference scenarios reset from the oracle's own generator + RNG stream.  The two generator-defined
    scenarios (spheres / noisy current) load the fixture's episode, after burning the draws the reference made
    on the shared stream (SimpleDocking3d: 1 + 3 + 3; NoisyCurrent: + 2) so the per-step normals stay aligned."""
    if env_name in orc.SCENARIOS:
        return env.reset(seed=seed)
    if seed is not None:
        env.rng = np.random.RandomState(seed)
    env.rng.random_sample()
    env.rng.random_sample(3)
    env.rng.random_sample(3)
    if env_name == "NoisyCurrentDocking3d":
        env.rng.random_sample(2)
    return env.reset(episode=episode_from_golden(g, e))


@pytest.mark.parametrize("name", TRAJ)
def test_g7_trajectory_free_running(name):
    """Free-running: the oracle is only given the seed and the action sequence.  For the seven reference scenarios
    the reset draws (G8) must come out of the oracle's own generator + RNG stream; for the two generator-defined
    scenarios (spheres / noisy current) episodes are loaded from the fixture (the RNG burn is still checked)."""
    g = load(name)
    env, env_name = env_from_meta(g)
    T = int(g["meta_T"])
    ep_start = g["ep_start"].tolist()
    reset_like_reference(env, g, 0, env_name, seed=int(g["meta_seed"]))
    e = 0
    for t in range(T):
        if t in ep_start and t > 0:
            e += 1
            reset_like_reference(env, g, e, env_name)
        if t in ep_start and "ep_pose_drawn" in g.files:
            # "near" fixtures: the generator moved the vehicle next to an obstacle after the reference's reset
            # (oracle/gen_golden.py: place_near_obstacle); the reset draw itself is still the oracle's own
            np.testing.assert_allclose(env.state[0:6], g["ep_pose_drawn"][e], rtol=1e-12, atol=1e-12)
            env.state[0:3] = g["ep_position"][e]
            env.state[3:6] = g["ep_attitude"][e]
        if t in ep_start:
            np.testing.assert_allclose(env.state[0:3], g["ep_position"][e], rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(env.state[3:6], g["ep_attitude"][e], rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(env.goal, g["ep_goal"][e], rtol=1e-12, atol=1e-12)
            assert env.heading_goal == pytest.approx(float(g["ep_heading_goal"][e]), abs=1e-12)
            cur = g["ep_current"][e]
            np.testing.assert_allclose([env.current.mu, env.current.V_min, env.current.V_max, env.current.V_c,
                                        env.current.alpha, env.current.beta, env.current.sigma], cur, atol=1e-12)
            n = int(g["ep_n_capsules"][e])
            assert len(env.capsules) == n
            for c, ref in zip(env.capsules, g["ep_capsules"][e][:n]):
                np.testing.assert_allclose(np.concatenate([c[0], c[1], [c[2]]]), ref, atol=1e-12)
            if g["ep_sph_radii"].shape[1] > 0:    # SphereDocking3d: the oracle's own obstacle-field generator
                np.testing.assert_allclose(env.sphere_centers, g["ep_sph_centers"][e], atol=1e-12)
                np.testing.assert_allclose(env.sphere_radii, g["ep_sph_radii"][e], atol=1e-12)
        obs, rew, done, _ = env.step(g["action"][t])
        ctx = f"{name} t={t}"
        np.testing.assert_allclose(env.state, g["state"][t], rtol=1e-9, atol=1e-9, err_msg=ctx)
        np.testing.assert_allclose(env.u, g["u"][t], rtol=1e-12, atol=1e-12, err_msg=ctx)
        np.testing.assert_allclose(env.nu_c, g["nu_c"][t], rtol=1e-9, atol=1e-10, err_msg=ctx)
        assert env.current.V_c == pytest.approx(float(g["V_c"][t]), abs=1e-12)
        np.testing.assert_allclose(env.state_dot[3:6], g["euler_dot"][t], rtol=1e-8, atol=1e-9, err_msg=ctx)
        np.testing.assert_allclose(env.intersec_dist, g["ray_dist"][t], rtol=1e-8, atol=1e-8, err_msg=ctx)
        np.testing.assert_allclose([env.delta_d, env.delta_theta, env.delta_psi, env.delta_heading_goal], g["nav"][t],
                                   rtol=1e-9, atol=1e-9, err_msg=ctx)
        np.testing.assert_allclose(obs, g["obs"][t], rtol=0, atol=2e-7, err_msg=ctx)
        np.testing.assert_allclose(env.last_reward_arr, g["reward_arr"][t], rtol=1e-8, atol=1e-9, err_msg=ctx)
        assert rew == pytest.approx(float(g["reward"][t]), rel=1e-9, abs=1e-9), ctx
        assert env.conditions == g["conditions"][t].tolist(), ctx
        assert done == bool(g["done"][t]) and env.collision == bool(g["collision"][t]), ctx
        assert env.t_steps == int(g["t_steps"][t]), ctx
