"""
GPU parity tests (run with -m gpu on the MI355X box): the HIP step kernel, called through the C ABI, against the golden
vectors generated from the reference (tests/golden, oracle/gen_golden.py).

Teacher-forced: env t of a batch starts from the state golden step t started from, gets golden action t, and must
reproduce golden step t -- every step of every trajectory is an independent check, one kernel launch per trajectory.
Tolerances: the float64 instantiation must agree to ~1e-9 (same algorithm, different summation order); the float32
product path within 1e-5 on observations (BASELINE.json north_star) and a relative 2e-5 on rewards -- teacher-forced AND
free-running (test_free_running_f32).
"""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu

def run_teacher_forced(name, precision):
    g = H.load(name)
    T = int(g["meta_T"])
    env, max_caps, max_sph = H.make_batched(g, T, precision, auto_reset=False)
    try:
        inp = H.teacher_forced_inputs(g, np.arange(T), max_caps, max_sph)
        H.load_teacher_forced(env, inp)
        obs, rew, done, infos = env.step(inp["actions"][:, :env.n_u], noise=inp["noise"], extras=True)
        return H.check_teacher_forced(env, obs, rew, done, inp["gold"], precision, name)
    finally:
        env.close()


@pytest.mark.parametrize("name", H.TRAJ)
def test_teacher_forced_f64(name):
    run_teacher_forced(name, "f64")


@pytest.mark.parametrize("name", H.TRAJ)
def test_teacher_forced_f32(name):
    run_teacher_forced(name, "f32")


def run_free(name, precision, tol_obs, tol_rayobs, tol_rew):
    """Free-running: only actions and episodes are given; the state is carried by the kernel across all steps
    (hundreds of steps, several episodes).  Steps where a ray flips hit/miss at grazing incidence (|dd| > 1 mm;
    unbounded condition number) are excluded from the comparison of the ray cells and the reward only -- obs[:16] is
    checked on every step -- and their number is bounded.  obs[2] =
    delta_psi / pi lives on a circle (ssa wraps it at +-1): it is compared modulo 2, so steps at the wrap are checked
    like any other (the reward term of delta_psi is even and continuous there)."""
    g = H.load(name)
    T = int(g["meta_T"])
    n_u = int(g["meta_n_u"])
    env, max_caps, max_sph = H.make_batched(g, 1, precision, auto_reset=False)
    try:
        _, _, _, _, w = H.prestep_inputs(g)
        ep_start = g["ep_start"].tolist()
        worst_obs = worst_ray = worst_rew = 0.0
        flips = 0
        e = -1
        for t in range(T):
            if t in ep_start:
                e += 1
                env.reset_envs([0], H.episode_arrays(g, [e], max_caps, max_sph))
            a = np.zeros((1, env.n_u))
            a[0, :n_u] = g["action"][t]
            obs, rew, done, _ = env.step(a, noise=w[t:t + 1], extras=True)
            assert bool(done[0]) == bool(g["done"][t]), f"{name}: done differs at step {t}"
            d = np.abs(obs[0, :16] - g["obs"][t, :16])
            d[2] = min(d[2], 2.0 - d[2])
            worst_obs = max(worst_obs, float(d.max()))      # (obs[:16] do not depend on the rays: every step counts)
            flip = bool((np.abs(env.intersec_dist[0] - g["ray_dist"][t]) > 1e-3).any())
            flips += flip
            if flip:
                continue
            worst_ray = max(worst_ray, float(np.abs(obs[0, 16:] - g["obs"][t, 16:]).max()))
            worst_rew = max(worst_rew, float(abs(rew[0] - g["reward"][t]) / max(1.0, abs(g["reward"][t]))))
        assert flips <= max(2, T // 100), f"{name}: {flips} steps with a flipped ray"
        assert worst_obs <= tol_obs, f"{name}: max |obs[:16] - ref| = {worst_obs}"
        assert worst_ray <= tol_rayobs, f"{name}: max |obs[16:] - ref| = {worst_ray}"
        assert worst_rew <= tol_rew, f"{name}: max rel reward error = {worst_rew}"
    finally:
        env.close()


@pytest.mark.parametrize("name", H.TRAJ)
def test_free_running_f32(name):
    # BASELINE.json north_star, literally: float32 within 1e-5 of the float64 reference step for step on identical seeds /
    # actions -- FREE-RUNNING over whole multi-episode trajectories (up to 450 steps; nothing pulls the float32 state back
    # to the reference's).  Measured (profiles/r3/parity_drift_f32.txt): obs[:16] <= 5.3e-6 on all 22 trajectories, reward
    # <= 1.5e-6 relative, ray cells <= 2.5e-5.  Round 3 found what round 2 had put down to "float32 itself": the angle
    # wrap ((a + pi) - pi costs 2.4e-7 rad per step in float32 even when nothing wraps; the undamped heading random-walked
    # and the position integrated it) -- now exact for angles in range -- plus the heading carried in two floats like the
    # position.  The bound for the ray cells is what the REFERENCE does to itself when its state is merely stored in
    # float32 (all arithmetic float64): up to 3.3e-5 on a cell at grazing incidence, 4.3e-6 on obs[:16]
    # (oracle/ulp_perturb_reference.py -> profiles/r3/reference_f32_storage_drift.txt).
    run_free(name, "f32", 1e-5, 5e-5, 1e-5)


@pytest.mark.parametrize("name", [n for n in H.TRAJ if "config1" in n or "Obstacles" in n])
def test_free_running_f64(name):
    run_free(name, "f64", 3e-7, 3e-7, 1e-8)
