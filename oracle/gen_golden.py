#!/usr/bin/env python3
"""
oracle/gen_golden.py -- generate the golden vectors under tests/golden/ by IMPORTING the reference
(/root/reference, present only in the build container; never on the GPU box).

TEST INFRASTRUCTURE ONLY.  Run:  python oracle/gen_golden.py            (re-creates tests/golden/*.npz)

The reference needs two third-party packages that are not installed here (``gym`` ~0.21 and
``skimage.measure.block_reduce``); both are replaced by minimal in-memory stand-ins *inside this script
only* (SURVEY.md section 8c).  ``block_reduce`` follows scikit-image's documented behaviour (pad with
cval=0 to a multiple of the block, then ``func`` per block); no reference test pins that boundary, so
the reduced ray image is pinned through this stand-in only (and the un-reduced ray distances are stored
as well, which do not depend on it).

Fixtures are DATA: inputs and the reference's outputs.  No reference source text is stored.
"""
import copy
import os
import sys
import tempfile
import types

import numpy as np

REF = os.environ.get("DOCKAUV_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def install_standins():
    sys.dont_write_bytecode = True
    import matplotlib
    matplotlib.use("Agg")

    gym = types.ModuleType("gym")

    class Env:
        def __init__(self):
            pass

    class Box:
        def __init__(self, low, high, dtype=np.float32):
            self.low = np.asarray(low, dtype=dtype)
            self.high = np.asarray(high, dtype=dtype)
            self.dtype = dtype
            self.shape = self.low.shape

    gym.Env = Env
    spaces = types.ModuleType("gym.spaces")
    spaces.Box = Box
    utils = types.ModuleType("gym.utils")
    seeding = types.ModuleType("gym.utils.seeding")
    seeding.np_random = lambda seed=None: (np.random.RandomState(seed), seed)
    envs = types.ModuleType("gym.envs")
    reg = types.ModuleType("gym.envs.registration")
    reg.register = lambda **kw: None
    gym.spaces, gym.utils, gym.envs = spaces, utils, envs
    utils.seeding, envs.registration = seeding, reg
    for name, mod in [("gym", gym), ("gym.spaces", spaces), ("gym.utils", utils), ("gym.utils.seeding", seeding),
                      ("gym.envs", envs), ("gym.envs.registration", reg)]:
        sys.modules[name] = mod

    sk = types.ModuleType("skimage")
    skm = types.ModuleType("skimage.measure")

    def block_reduce(image, block_size, func=np.sum, cval=0):
        b = block_size if isinstance(block_size, int) else block_size[0]
        r, c = image.shape
        R, C = -(-r // b) * b, -(-c // b) * b
        pad = np.full((R, C), cval, dtype=image.dtype)
        pad[:r, :c] = image
        return func(pad.reshape(R // b, b, C // b, b), axis=(1, 3))

    skm.block_reduce = block_reduce
    sk.measure = skm
    sys.modules["skimage"] = sk
    sys.modules["skimage.measure"] = skm
    sys.path.insert(0, REF)


install_standins()
from gym_dockauv.config.env_config import BASE_CONFIG  # noqa: E402
import gym_dockauv.envs.docking3d as d3  # noqa: E402
from gym_dockauv.objects import shape  # noqa: E402
from gym_dockauv.objects.current import Current  # noqa: E402
from gym_dockauv.objects.shape import Sphere, Spheres  # noqa: E402
from gym_dockauv.objects.vehicles.BlueROV2 import BlueROV2  # noqa: E402
from gym_dockauv.objects.vehicles.LAUV import LAUV  # noqa: E402
from gym_dockauv.utils import geomutils as geom  # noqa: E402

TMP = tempfile.mkdtemp(prefix="dockauv_golden_")


def make_cfg(**over):
    cfg = copy.deepcopy(BASE_CONFIG)
    cfg["save_path_folder"] = TMP
    cfg["verbose"] = 0
    cfg["log_level"] = 50
    cfg["interval_datastorage"] = 10 ** 9
    for k, v in over.items():
        if isinstance(v, dict):
            cfg[k].update(v)
        else:
            cfg[k] = v
    return cfg


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {os.path.relpath(path)}  ({os.path.getsize(path) / 1024:.1f} KiB)")


# ------------------------------------------------------------------------------------------------
# G1 / G2 / G3: vehicle constants, state_dot known answers, AUVSim.step transitions
# ------------------------------------------------------------------------------------------------

def vehicles():
    test_xml = os.path.join(REF, "gym_dockauv", "tests", "objects", "test_BlueROV2.xml")
    return {
        "bluerov2": BlueROV2(),
        "bluerov2_direct": BlueROV2(control_mode="direct"),
        "bluerov2_testxml": BlueROV2(test_xml),
        "lauv": LAUV(),
    }


def gen_constants():
    out = {}
    for name, v in vehicles().items():
        v.step_size = 0.1
        nu0 = np.array([1.3, -0.4, 0.2, 0.1, -0.2, 0.3])
        out[name + "_M_RB"] = v.M_RB
        out[name + "_M_A"] = v.M_A
        out[name + "_M_inv"] = v.M_inv
        out[name + "_I_b"] = v.I_b
        out[name + "_W_BY"] = np.array([v.W, v.BY])
        out[name + "_B_at_nu0"] = np.asarray(v.B(nu0), dtype=float)
        out[name + "_u_bound"] = np.asarray(v.u_bound, dtype=float)
        out[name + "_alpha_h0p1"] = np.array([v.lowpassfilter.alpha])
        out[name + "_C_at_nu0"] = v.C(nu0)
        out[name + "_D_at_nu0"] = v.D(nu0)
        out[name + "_G_at_eta0"] = v.G(np.array([0, 0, 0, 0.3, -0.2, 1.0]))
    out["nu0"] = np.array([1.3, -0.4, 0.2, 0.1, -0.2, 0.3])
    save("g1_constants", **out)


def gen_state_dot():
    rs = np.random.RandomState(2024)
    out = {}
    for name, v in vehicles().items():
        n_u = v.u_bound.shape[0]
        K = 96
        states = np.zeros((K, 12))
        states[:, 0:3] = rs.uniform(-20, 20, (K, 3))
        states[:, 3:5] = rs.uniform(-1.0, 1.0, (K, 2))
        states[:, 5] = rs.uniform(-np.pi, np.pi, K)
        states[:, 6:9] = rs.uniform(-2, 2, (K, 3))
        states[:, 9:12] = rs.uniform(-1.5, 1.5, (K, 3))
        lo, hi = v.u_bound[:, 0], v.u_bound[:, 1]
        us = lo + (hi - lo) * rs.uniform(0, 1, (K, n_u))
        nucs = np.zeros((K, 6))
        nucs[:, 0:3] = rs.uniform(-1, 1, (K, 3))
        nucs[: K // 4] = 0.0
        sd = np.zeros((K, 12))
        for i in range(K):
            v.u = us[i]
            sd[i] = v.state_dot(0, states[i], nucs[i])
        out[name + "_state"] = states
        out[name + "_u"] = us
        out[name + "_nu_c"] = nucs
        out[name + "_state_dot"] = sd
    save("g2_state_dot", **out)


def gen_auv_step():
    rs = np.random.RandomState(77)
    out = {}
    for name, v in vehicles().items():
        n_u = v.u_bound.shape[0]
        hs = [0.1, 0.05, 0.01] if not name.startswith("lauv") else [0.02, 0.01]
        for h in hs:
            # a fresh instance per h: the low-pass alpha is a cached_property frozen at first use (Q5)
            vv = vehicles()[name]
            vv.step_size = h
            K = 48
            states = np.zeros((K, 12))
            states[:, 0:3] = rs.uniform(-20, 20, (K, 3))
            states[:, 3:5] = rs.uniform(-0.9, 0.9, (K, 2))
            states[:, 5] = rs.uniform(-np.pi, np.pi, K)
            vel_scale = 1.0 if not name.startswith("lauv") else 0.5
            states[:, 6:9] = rs.uniform(-1.5, 1.5, (K, 3)) * vel_scale
            states[:, 9:12] = rs.uniform(-1.0, 1.0, (K, 3)) * vel_scale
            lo, hi = vv.u_bound[:, 0], vv.u_bound[:, 1]
            u_prev = lo + (hi - lo) * rs.uniform(0, 1, (K, n_u))
            actions = rs.uniform(-1.3, 1.3, (K, n_u))  # beyond [-1,1]: exercises the clip
            nucs = np.zeros((K, 6))
            nucs[:, 0:3] = rs.uniform(-0.5, 0.5, (K, 3))
            new_state = np.zeros((K, 12))
            new_u = np.zeros((K, n_u))
            new_sd = np.zeros((K, 12))
            for i in range(K):
                vv.state = states[i].copy()
                vv.u = u_prev[i].copy()
                vv.step(actions[i], nucs[i])
                new_state[i], new_u[i], new_sd[i] = vv.state, vv.u, vv._state_dot
            tag = f"{name}_h{h}"
            out[tag + "_state"] = states
            out[tag + "_u_prev"] = u_prev
            out[tag + "_action"] = actions
            out[tag + "_nu_c"] = nucs
            out[tag + "_new_state"] = new_state
            out[tag + "_new_u"] = new_u
            out[tag + "_new_state_dot"] = new_sd
    # the reference's own integrator test (tests/objects/test_BlueROV2.py:150-188): 100 steps, h = 0.01
    v = vehicles()["bluerov2_testxml"]
    v.set_B(np.identity(6))
    v.set_u_bound(np.array([[-5, 5], [-5, 5], [-5, 5], [-1, 3], [-1, 1], [-1, 1]], dtype=float))
    v.step_size = 0.01
    v.state = np.zeros(12)
    action = np.array([1, 0, 0, -0.5, 0, 0], dtype=float)
    for _ in range(100):
        v.step(action, np.zeros(6))
    out["test_sim_ode_final_state"] = v.state.copy()
    out["test_sim_ode_final_u"] = v.u.copy()
    save("g3_auv_step", **out)


# ------------------------------------------------------------------------------------------------
# G4 / G5: ray-obstacle kernels and collisions
# ------------------------------------------------------------------------------------------------

def gen_rays():
    rs = np.random.RandomState(5)
    out = {}
    # random capsules x random rays (+ edge cases appended)
    n_caps, n_rays = 12, 64
    cap1 = rs.uniform(-8, 8, (n_caps, 3))
    cap2 = cap1 + rs.uniform(-6, 6, (n_caps, 3))
    rad = rs.uniform(0.3, 2.0, n_caps)
    # make the first 4 capsules vertical like the shipped scenarios
    cap2[:4, 0:2] = cap1[:4, 0:2]
    origins = rs.uniform(-10, 10, (n_caps, n_rays, 3))
    dirs = rs.normal(size=(n_caps, n_rays, 3))
    # aim half of the rays roughly at the capsule so that hits are common
    mid = 0.5 * (cap1 + cap2)
    aim = mid[:, None, :] - origins + rs.normal(scale=1.0, size=origins.shape)
    dirs[:, ::2] = aim[:, ::2]
    # some origins inside the capsule
    origins[:, 5] = mid + 0.1
    dist = np.zeros((n_caps, n_rays))
    for i in range(n_caps):
        with np.errstate(all="ignore"):
            dist[i] = shape.intersec_dist_line_capsule_vectorized(origins[i], dirs[i], cap1[i], cap2[i], rad[i])
    out.update(cap_cap1=cap1, cap_cap2=cap2, cap_rad=rad, cap_origins=origins, cap_dirs=dirs, cap_dist=dist)

    # hand-made edge cases, one capsule z-axis [-2, 2], r = 1
    e_c1, e_c2, e_r = np.array([0., 0., 2.]), np.array([0., 0., -2.]), 1.0
    e_o = np.array([
        [5, 0, 0], [5, 0, 0], [5, 0, 0], [0, 0, 6], [0, 0, -6], [0.2, 0.1, 0.0], [5, 0, 2.0], [5, 0, 0],
        [5, 1.0, 0], [0, 0, 6], [3, 0, 3], [5, 0, 2.5], [5, 0, -2.5], [0, 5, 1.999],
    ], dtype=float)
    e_d = np.array([
        [-1, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, -1], [0, 0, 1], [1, 0, 0], [-1, 0, 0], [-1, 0, 0.5],
        [-1, 0, 0], [0.001, 0, -1], [-1, 0, -1], [-1, 0, 0], [-1, 0, 0], [0, -1, 0],
    ], dtype=float)
    with np.errstate(all="ignore"):
        e_dist = shape.intersec_dist_line_capsule_vectorized(e_o, e_d, e_c1, e_c2, e_r)
    out.update(edge_cap1=e_c1, edge_cap2=e_c2, edge_rad=np.array([e_r]), edge_origins=e_o, edge_dirs=e_d,
               edge_dist=e_dist)

    # spheres: random sets
    n_sets, n_sph = 10, 8
    centers = rs.uniform(-10, 10, (n_sets, n_sph, 3))
    radii = rs.uniform(0.5, 1.5, (n_sets, n_sph))
    s_o = rs.uniform(-12, 12, (n_sets, n_rays, 3))
    s_d = rs.normal(size=(n_sets, n_rays, 3))
    pick = rs.randint(0, n_sph, (n_sets, n_rays))
    for i in range(n_sets):
        tgt = centers[i, pick[i]] + rs.normal(scale=0.7, size=(n_rays, 3))
        s_d[i, ::2] = (tgt - s_o[i])[::2]
        s_o[i, 3] = centers[i, 0] + 0.05  # inside sphere 0
    s_dist = np.zeros((n_sets, n_rays))
    for i in range(n_sets):
        with np.errstate(all="ignore"):
            s_dist[i] = shape.intersec_dist_lines_spheres_vectorized(s_o[i], s_d[i], centers[i], radii[i])
    out.update(sph_centers=centers, sph_radii=radii, sph_origins=s_o, sph_dirs=s_d, sph_dist=s_dist)
    save("g4_rays", **out)


def gen_collision():
    rs = np.random.RandomState(9)
    K = 200
    pos = rs.uniform(-6, 6, (K, 3))
    cap1 = rs.uniform(-5, 5, (K, 3))
    cap2 = cap1 + rs.uniform(-5, 5, (K, 3))
    rad = rs.uniform(0.5, 1.5, K)
    dist = np.array([shape.dist_line_point(pos[i], cap1[i], cap2[i]) for i in range(K)])
    hit_c = np.array([shape.collision_capsule_sphere(cap1[i], cap2[i], rad[i], pos[i], 1.0) for i in range(K)])
    centers = rs.uniform(-6, 6, (K, 8, 3))
    radii = rs.uniform(0.5, 1.5, (K, 8))
    hit_s = np.array([shape.collision_sphere_spheres(pos[i], 1.0, centers[i], radii[i]) for i in range(K)])
    vec = np.array([shape.vec_line_point(pos[i], cap1[i], cap2[i]) for i in range(K)])
    save("g5_collision", pos=pos, cap1=cap1, cap2=cap2, rad=rad, seg_dist=dist, hit_capsule=hit_c,
         sph_centers=centers, sph_radii=radii, hit_spheres=hit_s, vec_line_point=vec)


# ------------------------------------------------------------------------------------------------
# G6 / G7 / G8: env-level trajectories (reset draws, states, observations, rewards, conditions)
# ------------------------------------------------------------------------------------------------

class SphereDocking3d(d3.SimpleDocking3d):
    """Build-defined scenario for BASELINE config 3: SimpleDocking3d + 8 sphere obstacles in a 3..12 m shell
    around the goal (the reference ships the sphere ray/collision functions but no env that uses them)."""
    sphere_seed = 10_000

    def generate_environment(self):
        super().generate_environment()
        rs = np.random.RandomState(self.sphere_seed + self.episode)
        d = rs.normal(size=(8, 3))
        d /= np.linalg.norm(d, axis=1)[:, None]
        centers = self.goal_location + d * rs.uniform(3.0, 12.0, 8)[:, None]
        radii = rs.uniform(0.5, 1.5, 8)
        self.spheres = Spheres([Sphere(c, r) for c, r in zip(centers, radii)])
        self.obstacles = [*self.spheres()]


class NoisyCurrentDocking3d(d3.SimpleDocking3d):
    """SimpleDocking3d with a Gauss-Markov current that actually moves (sigma > 0), as the reference's
    integration test builds it (tests/test_integration.py)."""

    def generate_environment(self):
        super().generate_environment()
        ang = (np.random.random(2) - 0.5) * 2 * np.array([np.pi / 2, np.pi])
        self.current = Current(mu=0.01, V_min=0.2, V_max=1.0, Vc_init=0.5, alpha_init=ang[0], beta_init=ang[1],
                               white_noise_std=0.1, step_size=self.auv.step_size)
        self.nu_c = self.current(self.auv.attitude)


ENV_CLASSES = {
    "SimpleDocking3d": d3.SimpleDocking3d,
    "SimpleCurrentDocking3d": d3.SimpleCurrentDocking3d,
    "CapsuleDocking3d": d3.CapsuleDocking3d,
    "CapsuleCurrentDocking3d": d3.CapsuleCurrentDocking3d,
    "ObstaclesDocking3d": d3.ObstaclesDocking3d,
    "ObstaclesNoCapDocking3d": d3.ObstaclesNoCapDocking3d,
    "ObstaclesCurrentDocking3d": d3.ObstaclesCurrentDocking3d,
    "SphereDocking3d": SphereDocking3d,
    "NoisyCurrentDocking3d": NoisyCurrentDocking3d,
}


def ctrl_random(rs, scale=1.0):
    def f(env):
        return rs.uniform(-1, 1, env.auv.u_bound.shape[0]) * scale
    return f


def ctrl_goto(rs, noise=0.1):
    """Crude heading / pitch / surge controller that drives towards the goal (BlueROV2 joystick or LAUV)."""
    def f(env):
        n_u = env.auv.u_bound.shape[0]
        diff = env.goal_location - env.auv.position
        d = np.linalg.norm(diff)
        dpsi = geom.ssa(np.arctan2(diff[1], diff[0]) - env.auv.attitude[2])
        dth = env.auv.attitude[1] + geom.ssa(np.arctan2(diff[2], np.linalg.norm(diff[:2])))
        a = np.zeros(n_u)
        if n_u == 6:
            a[0] = np.clip(0.9 * np.cos(dpsi), -1, 1) * min(1.0, d / 2.0 + 0.3)
            a[2] = np.clip(diff[2] * 0.8, -1, 1)
            a[5] = np.clip(dpsi * 1.5 - 0.8 * env.auv.angular_velocity[2], -1, 1)
            a[3] = np.clip(-env.auv.attitude[0] * 2.0, -1, 1)
            a[4] = np.clip(-env.auv.attitude[1] * 2.0, -1, 1)
        elif n_u == 3:
            a[0] = 0.2
            a[1] = np.clip(-dpsi * 1.0, -1, 1)
            a[2] = np.clip(dth * 1.0, -1, 1)
        else:
            a[:] = rs.uniform(-1, 1, n_u)
        return a + rs.normal(scale=noise, size=n_u)
    return f


def ctrl_flee(rs):
    def f(env):
        n_u = env.auv.u_bound.shape[0]
        diff = env.goal_location - env.auv.position
        dpsi = geom.ssa(np.arctan2(-diff[1], -diff[0]) - env.auv.attitude[2])
        a = np.zeros(n_u)
        if n_u == 6:
            a[0] = 1.0 * np.cos(dpsi)
            a[5] = np.clip(dpsi * 1.5, -1, 1)
        else:
            a[0] = 1.0
        return a + rs.normal(scale=0.05, size=n_u)
    return f


def ctrl_roll(rs):
    def f(env):
        n_u = env.auv.u_bound.shape[0]
        a = rs.normal(scale=0.05, size=n_u)
        if n_u >= 6:
            a[3] = 1.0
            a[4] = 0.6
        return a
    return f


def capsule_array(env):
    if len(env.capsules) == 0:
        return np.zeros((0, 7))
    return np.array([[*c.vec_bot, *c.vec_top, c.radius] for c in env.capsules], dtype=float)


def episode_record(env):
    c = env.current
    return dict(
        position=env.auv.position.copy(), attitude=env.auv.attitude.copy(), goal=np.array(env.goal_location, float),
        heading_goal=float(env.heading_goal_reached),
        current=np.array([c.mu, c.V_min, c.V_max, c.V_c, c.alpha, c.beta, c.white_noise_std], float),
        capsules=capsule_array(env),
        sph_centers=np.array(env.spheres.position, float).reshape(-1, 3),
        sph_radii=np.array(env.spheres.radius, float).reshape(-1),
    )


def place_near_obstacle(rs, dist=(4.0, 6.0)):
    """Episode hook: after the reference's own reset, move the vehicle to 4-6 m from the axis of one of the env's
    capsules, roughly at the goal's depth and facing the capsule (+-0.25 rad), through the reference's own setters
    (objects/auvsim.py:174-195) -- where a docking policy spends its steps: the goal sits ON the centre capsule's
    safety surface (docking3d.py:868-876), so an approach has the fan full of hits.  Positions closer than 2.6 m to
    any capsule axis are redrawn (no collision at t = 0)."""
    def f(env):
        caps = env.capsules
        axes = [0.5 * (np.asarray(c.vec_top, float) + np.asarray(c.vec_bot, float))[:2] for c in caps]
        for _ in range(1000):
            k = rs.randint(len(caps))
            ang, d = rs.uniform(-np.pi, np.pi), rs.uniform(*dist)
            pos = np.array([axes[k][0] + d * np.cos(ang), axes[k][1] + d * np.sin(ang),
                            env.goal_location[2] + rs.uniform(-1.0, 1.0)])
            if min(np.linalg.norm(pos[:2] - a) for a in axes) >= 2.6 and np.linalg.norm(pos - env.goal_location) < 15.0:
                break
        else:
            raise RuntimeError("no free position found")
        heading = np.arctan2(axes[k][1] - pos[1], axes[k][0] - pos[0]) + rs.uniform(-0.25, 0.25)
        env.auv.position = pos
        env.auv.attitude = np.array([rs.uniform(-0.1, 0.1), rs.uniform(-0.15, 0.15), geom.ssa(heading)])
    return f


def run_traj(name, env_name, vehicle, seed, T, controller, cfg_over=None, act_seed=1, post_reset=None):
    cfg = make_cfg(vehicle=vehicle, **(cfg_over or {}))
    env = ENV_CLASSES[env_name](cfg)
    rs = np.random.RandomState(act_seed)
    ctrl = controller(rs)
    obs0 = env.reset(seed=seed)
    assert np.all(obs0 == 0)
    drawn = [np.concatenate([env.auv.position, env.auv.attitude])]   # the reference's own reset draw (G8)
    if post_reset:
        post_reset(env)
    n_u = env.auv.u_bound.shape[0]
    R = env.radar.n_rays
    rec = {k: [] for k in ("action", "state", "u", "nu_c", "V_c", "euler_dot", "obs", "reward", "reward_arr",
                           "conditions", "done", "collision", "nav", "ray_dist", "ep_index", "t_steps")}
    episodes = [episode_record(env)]
    ep_start = [0]
    n_goal = 0
    import io
    import contextlib
    for t in range(T):
        a = ctrl(env)
        with contextlib.redirect_stdout(io.StringIO()):  # the reference prints "Goal reached"
            obs, rew, done, info = env.step(a)
        rec["action"].append(np.array(a, float))
        rec["state"].append(env.auv.state.copy())
        rec["u"].append(env.auv.u.copy())
        rec["nu_c"].append(env.nu_c.copy())
        rec["V_c"].append(float(env.current.V_c))
        rec["euler_dot"].append(env.auv.euler_dot.copy())
        rec["obs"].append(obs.copy())
        rec["reward"].append(float(rew))
        rec["reward_arr"].append(env.last_reward_arr.copy())
        rec["conditions"].append(np.array(env.conditions, dtype=bool))
        rec["done"].append(bool(done))
        rec["collision"].append(bool(env.collision))
        rec["nav"].append(np.array([env.delta_d, env.delta_theta, env.delta_psi, env.delta_heading_goal]))
        rec["ray_dist"].append(np.array(env.radar.intersec_dist, float).copy())
        rec["ep_index"].append(len(episodes) - 1)
        rec["t_steps"].append(int(info["t_step"]))
        n_goal += int(env.conditions[0])
        if done and t + 1 < T:
            env.reset()                      # no seed: the global stream continues (burned by per-step normals)
            drawn.append(np.concatenate([env.auv.position, env.auv.attitude]))
            if post_reset:
                post_reset(env)
            episodes.append(episode_record(env))
            ep_start.append(t + 1)
    arrays = {k: np.array(v) for k, v in rec.items()}
    n_cap_max = max(e["capsules"].shape[0] for e in episodes)
    E = len(episodes)
    arrays["ep_start"] = np.array(ep_start)
    arrays["ep_position"] = np.array([e["position"] for e in episodes])
    arrays["ep_attitude"] = np.array([e["attitude"] for e in episodes])
    arrays["ep_goal"] = np.array([e["goal"] for e in episodes])
    arrays["ep_heading_goal"] = np.array([e["heading_goal"] for e in episodes])
    if post_reset:   # pose the reference drew before the hook moved the vehicle (ep_position / ep_attitude = after)
        arrays["ep_pose_drawn"] = np.array(drawn)
    arrays["ep_current"] = np.array([e["current"] for e in episodes])
    caps = np.zeros((E, n_cap_max, 7))
    ncap = np.zeros(E, dtype=np.int64)
    for i, e in enumerate(episodes):
        ncap[i] = e["capsules"].shape[0]
        caps[i, : ncap[i]] = e["capsules"]
    arrays["ep_capsules"] = caps
    arrays["ep_n_capsules"] = ncap
    arrays["ep_sph_centers"] = np.array([e["sph_centers"] for e in episodes])
    arrays["ep_sph_radii"] = np.array([e["sph_radii"] for e in episodes])
    meta = dict(env=env_name, vehicle=vehicle, seed=seed, T=T, n_u=n_u, n_rays=R, n_obs=env.n_observations,
                t_step_size=cfg["t_step_size"], max_timesteps=cfg["max_timesteps"], reward_set=cfg["reward_set"],
                radar_alpha=cfg["radar"]["alpha"], radar_beta=cfg["radar"]["beta"],
                radar_ray_per_deg=cfg["radar"]["ray_per_deg"], radar_max_dist=cfg["radar"]["max_dist"],
                action_reward_factors=np.asarray(cfg["action_reward_factors"], float))
    for k, v in meta.items():
        arrays["meta_" + k] = np.array(v)
    conds = arrays["conditions"].sum(axis=0)
    in_range = arrays["ray_dist"] < float(cfg["radar"]["max_dist"])
    print(f"  {name}: episodes={E} cond counts goal/out/att/maxt/col={conds.tolist()} "
          f"min_ray={arrays['ray_dist'].min():.3f} rays in range: {in_range.mean():.3f} of all rays, "
          f"{in_range.any(axis=1).mean():.3f} of the steps")
    save(name, **arrays)
    return arrays


def gen_trajectories():
    fan16 = {"alpha": 30 * np.pi / 180, "beta": 30 * np.pi / 180, "ray_per_deg": 10 * np.pi / 180}
    lauv = {"t_step_size": 0.02}
    # plumbing case of BASELINE config 1: seed 0, RandomState(123) uniform actions
    run_traj("traj_config1_simple_bluerov2", "SimpleDocking3d", "BlueROV2", 0, 300, lambda rs: ctrl_random(rs), act_seed=123)
    # every scenario, BlueROV2, goto controller (reaches goal / collides), several episodes
    for i, name in enumerate(ENV_CLASSES):
        if name in ("SphereDocking3d", "NoisyCurrentDocking3d"):
            continue
        run_traj(f"traj_{name}_bluerov2_goto", name, "BlueROV2", 11 + i, 260, ctrl_goto,
                 cfg_over={"max_timesteps": 180}, act_seed=3 + i)
    # random actions with obstacles + short episodes: max-t terminations and stream burn across resets
    run_traj("traj_ObstaclesCurrentDocking3d_bluerov2_random", "ObstaclesCurrentDocking3d", "BlueROV2", 5, 200,
             lambda rs: ctrl_random(rs), cfg_over={"max_timesteps": 40}, act_seed=8)
    # out of range and attitude terminations
    run_traj("traj_SimpleDocking3d_bluerov2_flee", "SimpleDocking3d", "BlueROV2", 21, 160, ctrl_flee, act_seed=4)
    run_traj("traj_SimpleDocking3d_bluerov2_roll", "SimpleDocking3d", "BlueROV2", 22, 120, ctrl_roll, act_seed=5)
    # reward set 2
    run_traj("traj_ObstaclesDocking3d_bluerov2_rewardset2", "ObstaclesDocking3d", "BlueROV2", 31, 200, ctrl_goto,
             cfg_over={"reward_set": 2, "max_timesteps": 150}, act_seed=6)
    # config 3: 16-beam fan + 8 spheres
    run_traj("traj_SphereDocking3d_bluerov2_fan16", "SphereDocking3d", "BlueROV2", 41, 260, ctrl_goto,
             cfg_over={"radar": fan16, "max_timesteps": 200}, act_seed=7)
    run_traj("traj_SphereDocking3d_bluerov2_fan16_random", "SphereDocking3d", "BlueROV2", 42, 150,
             lambda rs: ctrl_random(rs), cfg_over={"radar": fan16, "max_timesteps": 60}, act_seed=9)
    # Gauss-Markov current with sigma > 0 (pins the per-step normal draw order)
    run_traj("traj_NoisyCurrentDocking3d_bluerov2_random", "NoisyCurrentDocking3d", "BlueROV2", 51, 150,
             lambda rs: ctrl_random(rs, 0.7), cfg_over={"max_timesteps": 70}, act_seed=10)
    # LAUV at h = 0.02 (h = 0.1 diverges in the reference itself)
    run_traj("traj_SimpleDocking3d_lauv_random", "SimpleDocking3d", "LAUV", 61, 250, lambda rs: ctrl_random(rs),
             cfg_over=lauv, act_seed=12)
    run_traj("traj_ObstaclesDocking3d_lauv_goto", "ObstaclesDocking3d", "LAUV", 62, 400, ctrl_goto,
             cfg_over=dict(lauv, max_timesteps=300), act_seed=13)
    run_traj("traj_ObstaclesCurrentDocking3d_lauv_random", "ObstaclesCurrentDocking3d", "LAUV", 63, 200,
             lambda rs: ctrl_random(rs), cfg_over=dict(lauv, max_timesteps=90), act_seed=14)
    gen_mixed_partner()
    gen_near_obstacles()
    gen_lauv_collision()


def gen_mixed_partner():
    """BlueROV2 under the configuration of traj_ObstaclesCurrentDocking3d_lauv_random (h = 0.02, max_timesteps 90):
    its rows interleaved with the LAUV rows are BASELINE config 5 (mixed 50/50 batch) checked directly against
    the reference (tests/test_gpu_fullsize.py)."""
    run_traj("traj_ObstaclesCurrentDocking3d_bluerov2_h002_random", "ObstaclesCurrentDocking3d", "BlueROV2", 64, 200,
             lambda rs: ctrl_random(rs), cfg_over={"t_step_size": 0.02, "max_timesteps": 90}, act_seed=15)


def gen_near_obstacles():
    """Round 3: LAUV (config 4) and the mixed pair (config 5) with the fan full of hits -- the two LAUV obstacle
    trajectories above never have a ray in range (min_ray = 10.0).  Vehicles start 4-6 m from a capsule, facing it."""
    near = {"t_step_size": 0.02, "max_timesteps": 90}
    run_traj("traj_ObstaclesDocking3d_lauv_near", "ObstaclesDocking3d", "LAUV", 71, 450,
             lambda rs: ctrl_goto(rs, noise=0.3), cfg_over=dict(near, max_timesteps=220), act_seed=16,
             post_reset=place_near_obstacle(np.random.RandomState(171)))
    run_traj("traj_ObstaclesCurrentDocking3d_lauv_near", "ObstaclesCurrentDocking3d", "LAUV", 72, 240,
             lambda rs: ctrl_goto(rs, noise=0.3), cfg_over=near, act_seed=17,
             post_reset=place_near_obstacle(np.random.RandomState(172)))
    run_traj("traj_ObstaclesCurrentDocking3d_bluerov2_h002_near", "ObstaclesCurrentDocking3d", "BlueROV2", 73, 240,
             lambda rs: ctrl_goto(rs, noise=0.3), cfg_over=near, act_seed=18,
             post_reset=place_near_obstacle(np.random.RandomState(173)))


def ctrl_ram(rs):
    """full thrust straight ahead (LAUV: fins neutral), small noise: drives a vehicle placed in front of a capsule into it"""
    def f(env):
        n_u = env.auv.u_bound.shape[0]
        a = rs.normal(scale=0.05, size=n_u)
        a[0] = 1.0
        return a
    return f


def gen_lauv_collision():
    """Round 3: the LAUV kernels' collision condition with the fan full of hits -- the vehicle starts 2.8-3.2 m from a capsule
    axis (collision at <= 2 m), facing it, full thrust."""
    run_traj("traj_ObstaclesDocking3d_lauv_ram", "ObstaclesDocking3d", "LAUV", 81, 420, ctrl_ram,
             cfg_over={"t_step_size": 0.02, "max_timesteps": 400}, act_seed=19,
             post_reset=place_near_obstacle(np.random.RandomState(181), dist=(2.8, 3.2)))


def gen_radar_layout():
    """Ray fan layout for the default 63-ray and the 16-ray fan (sensor.py:43-87) + beta_oa weights."""
    out = {}
    for tag, (al, be, rp) in {"fan63": (60, 80, 10), "fan16": (30, 30, 10), "fan_test": (30, 20, 5)}.items():
        from gym_dockauv.objects.sensor import Radar
        r = Radar(eta=np.zeros(6), freq=1, alpha=al * np.pi / 180, beta=be * np.pi / 180,
                  ray_per_deg=rp * np.pi / 180, max_dist=10)
        out[tag + "_alpha"] = r.alpha
        out[tag + "_beta"] = r.beta
        out[tag + "_rd_b"] = r.rd_b
        out[tag + "_shape"] = np.array([r.n_vertical, r.n_horizontal, r.n_rays, r.n_rays_reduced])
        att = np.array([0.2, -0.3, 1.1])
        r.update(np.array([1, 2, 3, *att]))
        out[tag + "_rd_n_att"] = r.rd_n
        out[tag + "_att"] = att
        d = np.linspace(0.5, 12.0, r.n_rays)
        r.update_intersec(d.copy())
        out[tag + "_d_in"] = d
        out[tag + "_d_clamped"] = r.intersec_dist.copy()
        out[tag + "_d_reduced"] = r.intersec_dist_reduced.copy()
        out[tag + "_oa"] = np.array([d3.Reward.obstacle_avoidance(r.alpha, r.beta, r.intersec_dist, r.alpha_max,
                                                                  r.beta_max, r.max_dist, 1, 0.001, 0.01)])
    save("g4_radar_layout", **out)


if __name__ == "__main__":
    print("reference:", REF)
    if "--mixed-partner-only" in sys.argv:    # added in round 2; the other fixtures are unchanged
        gen_mixed_partner()
        sys.exit(0)
    if "--near-only" in sys.argv:             # added in round 3; the other fixtures are unchanged
        gen_near_obstacles()
        sys.exit(0)
    if "--ram-only" in sys.argv:              # added in round 3
        gen_lauv_collision()
        sys.exit(0)
    gen_constants()
    gen_state_dot()
    gen_auv_step()
    gen_rays()
    gen_collision()
    gen_radar_layout()
    gen_trajectories()
    print("done")
