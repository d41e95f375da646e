"""
Host-side scenario generation (what ``reset()`` decides): start pose, goal, current and obstacles of an episode,
vectorised over the envs that need a new episode.

Mirrors the reference's generators ``generate_random_pos / generate_random_att`` (envs/docking3d.py:687-703) and the
seven ``generate_environment`` scenarios (envs/docking3d.py:795-988).  The generators consume uniform [0, 1) draws in
exactly the reference's order, so feeding them the draws of a legacy ``np.random.RandomState(seed)`` stream reproduces
the reference's episode for that seed (parity mode); feeding them any other uniform source gives the same
distribution (throughput mode).
"""
from __future__ import annotations

from typing import Dict

import numpy as np

SCENARIOS = ("SimpleDocking3d", "SimpleCurrentDocking3d", "CapsuleDocking3d", "CapsuleCurrentDocking3d",
             "ObstaclesDocking3d", "ObstaclesNoCapDocking3d", "ObstaclesCurrentDocking3d", "SphereDocking3d")

# uniform draws per reset, in stream order
N_DRAWS = {
    "SimpleDocking3d": 7,              # heading(1) pos(3) att(3)
    "SimpleCurrentDocking3d": 10,      # + current angles(2) speed(1)
    "CapsuleDocking3d": 9,             # + theta(1) goal z(1)
    "CapsuleCurrentDocking3d": 11,     # + current angles(2)
    "ObstaclesDocking3d": 10,          # + ring angle(1)
    "ObstaclesNoCapDocking3d": 10,
    "ObstaclesCurrentDocking3d": 12,   # + current angles(2)
    "SphereDocking3d": 7,              # spheres come from their own per-env stream (BASELINE config 3)
}
N_CAPSULES = {"SimpleDocking3d": 0, "SimpleCurrentDocking3d": 0, "CapsuleDocking3d": 1, "CapsuleCurrentDocking3d": 1,
              "ObstaclesDocking3d": 5, "ObstaclesNoCapDocking3d": 4, "ObstaclesCurrentDocking3d": 5,
              "SphereDocking3d": 0}
N_SPHERES = {s: 0 for s in SCENARIOS}
N_SPHERES["SphereDocking3d"] = 8

DISTANCE_FROM_GOAL = 15.0       # docking3d.py:809
CAPSULE_RADIUS = 1.0            # docking3d.py:864
CAPSULE_HEIGHT = 4.0            # docking3d.py:865
PILLAR_RADIUS = 1.0             # docking3d.py:923
PILLAR_DISTANCE = 6.0           # docking3d.py:925
N_PILLARS = 4                   # docking3d.py:926
SAFETY_RADIUS = 1.0             # objects/auvsim.py:43
CURRENT_MU = 0.005              # docking3d.py:820


def ssa(a):
    return (a + np.pi) % (2 * np.pi) - np.pi


def episodes_from_uniforms(scenario: str, U: np.ndarray, max_attitude: float, max_dist_from_goal: float,
                           max_capsules: int, max_spheres: int) -> Dict[str, np.ndarray]:
    """
    U: [n, N_DRAWS[scenario]] uniforms in stream order.  Returns host arrays in the layouts of the C ABI fields:
    pose [n,6], goal [n,4], current [n,5] = (V_c, V_min, V_max, alpha, beta), capsules [n, max_capsules*7],
    spheres [n, max_spheres*4] (unused slots: radius -1).
    """
    if scenario not in SCENARIOS:
        raise KeyError(f"unknown scenario {scenario!r}")
    U = np.asarray(U, dtype=np.float64)
    n = U.shape[0]
    if U.shape[1] < N_DRAWS[scenario]:
        raise ValueError("not enough uniform draws")
    k = 0
    heading = (U[:, 0] - 0.5) * np.pi                                     # docking3d.py:814
    r = U[:, 1:4] - 0.5                                                   # :694
    r[:, 2] = np.abs(r[:, 0] + r[:, 1]) / 3 * np.sign(r[:, 2])            # :695
    pos = r * (DISTANCE_FROM_GOAL / np.linalg.norm(r, axis=1))[:, None]   # :696, goal is still (0,0,0)
    att = (U[:, 4:7] - 0.5) * 2 * np.array([max_attitude * 0.7, max_attitude * 0.7, np.pi])   # :699-703
    k = 7
    goal = np.zeros((n, 4))
    goal[:, 3] = heading
    current = np.zeros((n, 5))                                            # V_c V_min V_max alpha beta  (:820-822)
    capsules = np.zeros((n, max_capsules, 7))
    capsules[:, :, 6] = -1.0
    spheres = np.zeros((n, max_spheres, 4))
    spheres[:, :, 3] = -1.0

    def current_angles(kk):
        return (U[:, kk:kk + 2] - 0.5) * 2 * np.array([np.pi / 2, np.pi])

    if scenario == "SimpleCurrentDocking3d":                              # :844-848
        ang = current_angles(k)
        speed = U[:, k + 2] * 1.0
        current = np.stack([np.full(n, 0.5), speed, speed, ang[:, 0], ang[:, 1]], axis=1)
        k += 3
    if scenario in ("CapsuleDocking3d", "CapsuleCurrentDocking3d", "ObstaclesDocking3d", "ObstaclesNoCapDocking3d",
                    "ObstaclesCurrentDocking3d"):
        theta = U[:, k] * 2 * np.pi                                       # :871
        radius = CAPSULE_RADIUS + SAFETY_RADIUS
        goal[:, 0] = np.cos(theta) * radius
        goal[:, 1] = np.sin(theta) * radius
        goal[:, 2] = (U[:, k + 1] - 0.5) * CAPSULE_HEIGHT                 # :876
        k += 2
        # capsule at the origin: position 0, vec_top (0,0,-h/2), vec_bot = 2*pos - top  (:878-880, shape.py:98-108)
        caps = [np.tile(np.array([0, 0, CAPSULE_HEIGHT / 2, 0, 0, -CAPSULE_HEIGHT / 2, CAPSULE_RADIUS]), (n, 1))]
        # heading at goal: vector from goal to its projection on the capsule axis (:884-886, shape.py:420-433)
        goal[:, 3] = ssa(np.arctan2(-goal[:, 1], -goal[:, 0]))
        if scenario.startswith("Obstacles"):                              # :923-946
            half = 2 * max_dist_from_goal / 2.0
            th = U[:, k] * 2 * np.pi
            k += 1
            for _ in range(N_PILLARS):
                x, y = np.cos(th) * PILLAR_DISTANCE, np.sin(th) * PILLAR_DISTANCE
                th = th + 2 * np.pi / N_PILLARS
                caps.append(np.stack([x, y, np.full(n, half), x, y, np.full(n, -half), np.full(n, PILLAR_RADIUS)], axis=1))
            if scenario == "ObstaclesNoCapDocking3d":                     # :964
                caps.pop(0)
        if len(caps) > max_capsules:
            raise ValueError(f"{scenario} needs {len(caps)} capsule slots, only {max_capsules} configured")
        for c, arr in enumerate(caps):
            capsules[:, c, :] = arr
        if scenario in ("CapsuleCurrentDocking3d", "ObstaclesCurrentDocking3d"):   # :904-906, :984-986
            ang = current_angles(k)
            current = np.stack([np.full(n, 0.5), np.full(n, 0.5), np.full(n, 0.5), ang[:, 0], ang[:, 1]], axis=1)
            k += 2
    return {
        "pose": np.concatenate([pos, att], axis=1),
        "goal": goal,
        "current": current,
        "capsules": capsules.reshape(n, max_capsules * 7),
        "spheres": spheres.reshape(n, max_spheres * 4),
    }


def sphere_shell(rs: np.random.RandomState, goal: np.ndarray, n_spheres: int = 8, r_min: float = 3.0,
                 r_max: float = 12.0) -> np.ndarray:
    """Build-defined obstacle field of BASELINE config 3: centres uniform in direction, distance U(r_min, r_max) from
    the goal, radii U(0.5, 1.5).  Returns [n_spheres, 4]."""
    d = rs.normal(size=(n_spheres, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    centers = np.asarray(goal, dtype=float)[None, :3] + d * rs.uniform(r_min, r_max, n_spheres)[:, None]
    radii = rs.uniform(0.5, 1.5, n_spheres)
    return np.concatenate([centers, radii[:, None]], axis=1)
