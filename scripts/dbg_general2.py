import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gym_dockauv_amd.envs.batched import BatchedDocking3d
def run(force_general, threads, scenario):
    kw = dict(num_envs=200, scenario=scenario, precision="f64", reset_mode="none", rng="batched", threads_per_group=threads)
    if force_general:
        kw["_force_general"] = True
    env = BatchedDocking3d(**kw)
    env._gen = np.random.default_rng(5)
    env.reset()
    rs = np.random.RandomState(2)
    o, r, d, _ = env.step(rs.uniform(-1, 1, (200, 6)))
    st = env.state.copy()
    env.close()
    return o, st
for scenario in ("ObstaclesCurrentDocking3d", "SimpleDocking3d"):
    o0, s0 = run(False, 256, scenario)
    for fg, th in ((True, 256), (True, 64), (False, 64)):
        o, s = run(fg, th, scenario)
        ds = np.abs(s - s0)
        bad = np.where(ds.max(axis=1) > 1e-9)[0]
        print(scenario, "general" if fg else "sym", "threads", th, "state diff per component", np.array2string(ds.max(axis=0), precision=2), "bad envs", len(bad), bad[:20])
