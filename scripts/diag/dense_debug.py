import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from gym_dockauv_amd.envs.batched import BatchedDocking3d
from gym_dockauv_amd import _capi
wl = bench.workload(3, 256)
env = BatchedDocking3d(wl["cfg"], num_envs=256, scenario=wl["scenario"], device=0, precision="f32", reset_mode="none", rng="batched")
env._gen = np.random.default_rng(1)
env.reset()
rng = np.random.default_rng(2)
bench.place_ray_dense(env, rng)
sph = env.get_field(_capi.F_SPHERES).reshape(256, -1, 4)
st = env.get_field(_capi.F_STATE)
print("sphere0", sph[0, 0], "state", st[0, :6])
a = np.zeros((256, env.n_u)); a[:, 2] = 1.985 / 80
obs, rew, done, _ = env.step(a, extras=True)
print("cells", obs[:4, 16:], "ray_dist", env.intersec_dist[0])
print("active", (obs[:, 16:] < 1).any(axis=1).mean(), "any ray hit", (env.intersec_dist < 10).any(axis=1).mean())
