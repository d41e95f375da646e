#!/usr/bin/env python3
"""How many envs of the bench workloads see an obstacle at all (ray cells < 1) in the steady state of bench.py's
action stream: sizes the gain of handling obstacle-free envs outside the ray passes."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

for cid in (3, 4, 5):
    wl = bench.workload(cid, 0)
    env = bench.make_env(wl, 0, 0, 0)
    N, n_obs, n_u = wl["envs"], env.n_observations, env.n_u
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev); gen.manual_seed(1234)
    a = torch.rand((64, N, n_u), device=dev, generator=gen) * 2 - 1
    out = torch.zeros((N, n_obs + 2), device=dev)
    s = torch.cuda.current_stream().cuda_stream
    for t in range(3001):
        env.step_device(a[t % 64].data_ptr(), out.data_ptr(), stream=s, packed=True)
        if t in (0, 10, 100, 300, 1000, 3000):
            torch.cuda.synchronize()
            cells = out[:, 16:n_obs]
            hit = (cells < 1.0).any(dim=1).float().mean().item()
            done = (out[:, n_obs + 1] > 0.5).float().mean().item()
            print(f"config{cid} step {t}: envs with a ray hit {hit:.3f}, ray cells < 1: {(cells < 1.0).float().mean().item():.3f}, done this step {done:.4f}")
    env.close()
