#!/usr/bin/env python3
"""
oracle/ulp_perturb_reference.py -- how far does the REFERENCE drift from itself when its state is kept in float32?

TEST INFRASTRUCTURE ONLY (imports /root/reference like oracle/gen_golden.py; build container only).

BASELINE.json's north_star asks for float32 results "within 1e-5 step-for-step on identical seeds/actions".  A free-
running replay carries the state across hundreds of steps.  This script separates what float32 STORAGE of the state costs
from what float32 ARITHMETIC costs: it replays every golden trajectory (same seed, same recorded actions, resets at the
same steps) in the reference itself -- all arithmetic in float64, exactly the reference's code -- and after every step
rounds chosen words of the state to the nearest float32 (at most half a float32 ulp per word and step).  The divergence
of that replay from the unperturbed golden trajectory is a bound no float32 implementation with that storage format can
beat, whatever its arithmetic.

Modes (what is rounded to float32 after each step; the filtered input u and the current speed V_c are rounded in all of
them, as the product path stores them in one float):
  none      nothing (sanity: the replay must reproduce the fixture bit for bit)
  vel       linear / angular velocities                         (attitude and position kept in float64)
  att+vel   + Euler angles   = the HIP path's storage in round 2 (position carried in two floats)
  all       + position       = plain float32 state (round 1)

Output: per trajectory and mode the worst |obs[:16] - golden|, the worst ray-cell difference, the worst position /
attitude error.  `python oracle/ulp_perturb_reference.py > profiles/r3/reference_f32_storage_drift.txt`
"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import gen_golden as gg  # noqa: E402  (installs the gym / skimage stand-ins, imports the reference)

GOLDEN = gg.OUT
MODES = ("none", "vel", "att+vel", "all")


def f32(x):
    return np.asarray(x, dtype=np.float64).astype(np.float32).astype(np.float64)


def replay(name, mode):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    env_name, vehicle = str(g["meta_env"]), str(g["meta_vehicle"])
    over = {"t_step_size": float(g["meta_t_step_size"]), "max_timesteps": int(g["meta_max_timesteps"]),
            "reward_set": int(g["meta_reward_set"]),
            "radar": {"alpha": float(g["meta_radar_alpha"]), "beta": float(g["meta_radar_beta"]),
                      "ray_per_deg": float(g["meta_radar_ray_per_deg"]), "max_dist": float(g["meta_radar_max_dist"])}}
    env = gg.ENV_CLASSES[env_name](gg.make_cfg(vehicle=vehicle, **over))
    T = int(g["meta_T"])
    ep_start = g["ep_start"].tolist()
    placed = "ep_pose_drawn" in g.files

    def start_episode(e, seed=None):
        env.reset(seed=seed)
        if placed:   # "near" fixtures: the generator moved the vehicle after the reference's reset
            env.auv.position = g["ep_position"][e].copy()
            env.auv.attitude = g["ep_attitude"][e].copy()
        assert np.allclose(env.auv.position, g["ep_position"][e], atol=1e-12), (name, e)

    start_episode(0, int(g["meta_seed"]))
    e = 0
    worst = dict(obs16=0.0, cells=0.0, pos=0.0, att=0.0, rew=0.0)
    done_mismatch = 0
    for t in range(T):
        if t in ep_start and t > 0:
            e += 1
            start_episode(e)
        with contextlib.redirect_stdout(io.StringIO()):
            obs, rew, done, _ = env.step(g["action"][t])
        d = np.abs(obs.astype(np.float64) - g["obs"][t])
        d[2] = min(d[2], abs(2.0 - d[2]))
        flip = (np.abs(np.asarray(env.radar.intersec_dist) - g["ray_dist"][t]) > 1e-3).any()
        worst["obs16"] = max(worst["obs16"], float(d[:16].max()))
        if not flip:
            worst["cells"] = max(worst["cells"], float(d[16:].max()) if d.size > 16 else 0.0)
            worst["rew"] = max(worst["rew"], abs(float(rew) - float(g["reward"][t])) / max(1.0, abs(float(g["reward"][t]))))
        worst["pos"] = max(worst["pos"], float(np.abs(env.auv.state[0:3] - g["state"][t][0:3]).max()))
        da = np.abs(env.auv.state[3:6] - g["state"][t][3:6])
        worst["att"] = max(worst["att"], float(np.minimum(da, np.abs(2 * np.pi - da)).max()))
        done_mismatch += int(bool(done) != bool(g["done"][t]))
        # ---- the perturbation: float32 storage of the chosen words
        if mode != "none":
            st = env.auv.state.copy()
            st[6:12] = f32(st[6:12])
            if mode in ("att+vel", "all"):
                st[3:6] = f32(st[3:6])
            if mode == "all":
                st[0:3] = f32(st[0:3])
            env.auv.state = st
            env.auv.u = f32(env.auv.u)
            env.current.V_c = float(f32(env.current.V_c))
    worst["done_mismatch"] = done_mismatch
    return worst


def main():
    names = sorted(os.path.basename(p)[:-4] for p in os.listdir(GOLDEN) if p.startswith("traj_"))
    names = sorted(set(n[:-4] if n.endswith(".npz") else n for n in names))
    only = [a for a in sys.argv[1:] if not a.startswith("-")]
    if only:
        names = [n for n in names if any(o in n for o in only)]
    print(__doc__.split("Output:")[0].strip().split("\n\n", 1)[1])
    print()
    print(f"{'trajectory':58s} {'mode':8s} {'obs[:16]':>9s} {'cells':>9s} {'rew_rel':>9s} {'pos [m]':>9s} {'att [rad]':>9s} {'done!=':>6s}")
    summary = {m: 0.0 for m in MODES}
    for name in names:
        for mode in MODES:
            w = replay(name, mode)
            if mode == "none":
                assert w["obs16"] == 0.0 and w["pos"] == 0.0 and w["done_mismatch"] == 0, (name, w)
            summary[mode] = max(summary[mode], w["obs16"])
            print(f"{name:58s} {mode:8s} {w['obs16']:9.2e} {w['cells']:9.2e} {w['rew']:9.2e} {w['pos']:9.2e} {w['att']:9.2e} {w['done_mismatch']:6d}")
    print()
    for m in MODES:
        print(f"worst obs[:16] divergence over all trajectories, mode {m:8s}: {summary[m]:.2e}")


if __name__ == "__main__":
    main()
