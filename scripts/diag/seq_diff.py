#!/usr/bin/env python3
"""Where does the resident step sequence differ from single launches?  (diagnosis of tests/test_gpu_reset.py:
test_step_sequence_equals_single_steps)   usage: seq_diff.py <config id> <envs> <threads> [K]"""
import copy
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from gym_dockauv_amd.envs.batched import BatchedDocking3d

cid, N, threads = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
K = int(sys.argv[4]) if len(sys.argv) > 4 else 70
dev = torch.device("cuda", 0)
wl = bench.workload(cid, N)
cfg = copy.deepcopy(wl["cfg"])
cfg["max_timesteps"] = 23
outs = {}
for mode in ("single", "resident"):
    env = BatchedDocking3d(cfg, num_envs=N, scenario=wl["scenario"], precision="f32", reset_mode="device", device_seed=99,
                           rng="batched", vehicles=wl["vehicles"], threads_per_group=threads)
    env._gen = np.random.default_rng(3)
    env.reset()
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    acts = torch.rand((K, N, env.n_u), device=dev, generator=g) * 2 - 1
    out = torch.zeros((K, N, env.n_observations + 2), device=dev)
    s = torch.cuda.current_stream().cuda_stream
    if mode == "single":
        for k in range(K):
            env.step_device(acts[k].data_ptr(), out[k].data_ptr(), stream=s, packed=True)
    else:
        ios = env.make_step_sequence([acts[k].data_ptr() for k in range(K)], [out[k].data_ptr() for k in range(K)])
        env.run_step_sequence(ios, stream=s)
    torch.cuda.synchronize()
    outs[mode] = out.cpu().numpy()
    n_obs = env.n_observations
    env.close()
a, b = outs["single"], outs["resident"]
d = a.view(np.uint32) != b.view(np.uint32)
print("rows differing per step:", d.any(axis=2).sum(axis=1).tolist())
if d.any():
    k = int(np.flatnonzero(d.any(axis=(1, 2)))[0])
    envs = np.flatnonzero(d[k].any(axis=1))
    print(f"first differing step {k}: envs {envs[:20].tolist()} ({envs.size} of {N}); vehicles {[wl['vehicles'][i] if wl['vehicles'] else '-' for i in envs[:8]]}")
    for e in envs[:4]:
        cols = np.flatnonzero(d[k, e])
        print(f"  env {e}: columns {cols.tolist()} (n_obs {n_obs}); single {a[k, e, cols][:6]} resident {b[k, e, cols][:6]} |diff| {np.abs(a[k, e, cols] - b[k, e, cols])[:6]}")
        if k > 0:
            print(f"    done in step {k - 1}: single {a[k - 1, e, n_obs + 1]} resident {b[k - 1, e, n_obs + 1]}")
