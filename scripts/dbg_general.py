import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gym_dockauv_amd.envs.batched import BatchedDocking3d
for precision in ("f64", "f32"):
    outs = []
    for force_general in (False, True):
        kw = dict(num_envs=200, scenario="ObstaclesCurrentDocking3d", precision=precision, reset_mode="none", rng="batched")
        if force_general:
            kw["_force_general"] = True
        env = BatchedDocking3d(**kw)
        env._gen = np.random.default_rng(5)
        env.reset()
        rs = np.random.RandomState(2)
        traj = []
        for t in range(10):
            o, r, d, _ = env.step(rs.uniform(-1, 1, (200, 6)))
            traj.append((o.copy(), r.copy(), d.copy()))
        outs.append(traj)
        env.close()
    for t, ((o1, r1, d1), (o2, r2, d2)) in enumerate(zip(*outs)):
        diff = np.abs(o1 - o2)
        i, k = np.unravel_index(np.nanargmax(diff), diff.shape)
        print(precision, "step", t, "max diff", diff.max(), "env", i, "obs idx", k, o1[i, k], o2[i, k], "n envs differing", int((diff.max(axis=1) > 1e-4).sum()), "done", int(d1.sum()), int(d2.sum()))
