#!/usr/bin/env python3
"""f64: general (cross-product) kinetics against the structural fast path, step by step: where do they differ?
usage: [DOCKAUV_LIB=...] python scripts/diag/general_vs_sym.py [threads]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gym_dockauv_amd.envs.batched import BatchedDocking3d
from gym_dockauv_amd import _capi
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 0
scn = sys.argv[2] if len(sys.argv) > 2 else "ObstaclesCurrentDocking3d"
outs = []
for force_general in (False, True):
    env = BatchedDocking3d(num_envs=200, scenario=scn, precision="f64", reset_mode="none", rng="batched",
                           _force_general=force_general, threads_per_group=threads)
    env._gen = np.random.default_rng(5)
    env.reset()
    rs = np.random.RandomState(2)
    traj = []
    for t in range(10):
        o, r, d, _ = env.step(rs.uniform(-1, 1, (200, 6)), extras=True)
        traj.append((o.copy(), r.copy(), env.state.copy(), np.asarray(env.state_dot).copy(), np.asarray(env.nav_errors).copy()))
    outs.append(traj)
    env.close()
for t, (a, b) in enumerate(zip(*outs)):
    do = np.abs(a[0] - b[0])
    ds = np.abs(a[2] - b[2])
    bad_envs = np.flatnonzero(ds.max(axis=1) > 1e-9)
    print(f"step {t}: max|dobs| {do.max():.3e} max|dstate| {ds.max():.3e} bad envs {bad_envs[:12].tolist()} (n={bad_envs.size})"
          + (f" worst state cols {np.argsort(-ds.max(axis=0))[:4].tolist()} nan={np.isnan(b[2]).any()}" if bad_envs.size else ""))
    if bad_envs.size and t < 3:
        i = bad_envs[0]
        print("   env", i, "sym    ", np.array2string(a[2][i], precision=6))
        print("   env", i, "general", np.array2string(b[2][i], precision=6))
