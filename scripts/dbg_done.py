import numpy as np, sys
sys.path.insert(0, "/root/repo")
from gym_dockauv_amd.envs.batched import BatchedDocking3d
env = BatchedDocking3d(num_envs=4096, scenario="SimpleDocking3d", auto_reset=True, rng="batched")
env.reset()
rs = np.random.RandomState(0)
ring = rs.uniform(-1, 1, (16, 4096, 6))
tot = np.zeros(5, int)
for t in range(400):
    obs, rew, done, infos = env.step(ring[t % 16])
    c = env.conditions
    tot += c.sum(0)
    if t % 50 == 0 or t > 390:
        print(t, int(done.sum()), c.sum(0), float(np.abs(env.state[:, 3:5]).max()))
print("totals", tot)
