import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gym_dockauv_amd.envs.batched import BatchedDocking3d
N, K = 333, 40
for scenario, layouts in (("SimpleCurrentDocking3d", (64, 128)), ("ObstaclesCurrentDocking3d", (64, 256, 512))):
    outs = []
    for th in layouts:
        env = BatchedDocking3d(num_envs=N, scenario=scenario, precision="f32", reset_mode="device", device_seed=5, rng="batched", threads_per_group=th)
        env._gen = np.random.default_rng(4); env.reset()
        rs = np.random.RandomState(9); tr = []
        for k in range(K):
            o, r, d, infos = env.step(rs.uniform(-1, 1, (N, env.n_u)), extras=True)
            tr.append((o, r, d.astype(float), env.last_reward_arr.copy(), env.state.copy()))
        outs.append(tr); env.close()
    for li, other in enumerate(outs[1:]):
        for k, (a, b) in enumerate(zip(outs[0], other)):
            diffs = [float(np.nanmax(np.abs(x - y))) for x, y in zip(a, b)]
            if max(diffs) > 0:
                i = int(np.argmax(np.abs(a[0] - b[0]).max(axis=1)))
                print(scenario, layouts[li + 1], "step", k, "max diffs obs/rew/done/terms/state", diffs, "env", i, "done there", a[2][i], b[2][i])
                break
        else:
            print(scenario, layouts[li + 1], "identical over", K, "steps")
