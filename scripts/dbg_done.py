import numpy as np, sys
sys.path.insert(0, ".")
from gym_dockauv_amd.envs.batched import BatchedDocking3d
for mode in ("pool", "device"):
    env = BatchedDocking3d(num_envs=4096, scenario="SimpleDocking3d", reset_mode=mode, rng="batched", device_seed=77)
    env.reset()
    rs = np.random.RandomState(0)
    ring = rs.uniform(-1, 1, (64, 4096, 6))
    tot = np.zeros(5, int); nd = 0
    for t in range(600):
        obs, rew, done, infos = env.step(ring[t % 64])
        tot += env.conditions.sum(0); nd += done.sum()
        if t in (100, 300, 599):
            print(mode, t, int(done.sum()), env.conditions.sum(0), "tsteps mean", env.t_steps.mean())
    print(mode, "totals", tot, "done rate/step", nd / 600 / 4096)
    env.close()
