"""kernel us of the product kernels without / with the terminal-observation copy (TERM), BASELINE configs"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
dev = torch.device("cuda", 0)
for cid in (2, 3, 4, 5):
    wl = bench.workload(cid, 0)
    env = bench.make_env(wl, 0, 0, 0)
    N, n, nu = wl["envs"], env.n_observations, env.n_u
    a = torch.rand((16, N, nu), device=dev) * 2 - 1
    out = torch.zeros((N, n + 2), device=dev)
    term = torch.zeros((N, n), device=dev)
    s = torch.cuda.current_stream().cuda_stream
    for i in range(600):
        env.step_device(a[i % 16].data_ptr(), out.data_ptr(), stream=s, packed=True)
    res = {}
    for rep in range(3):
        for name, tp in (("plain", 0), ("term", term.data_ptr())):
            us = sum(env.time_steps_device(a[i % 16].data_ptr(), out.data_ptr(), steps=1, stream=s, packed=True, terminal_obs_ptr=tp) for i in range(400)) / 400
            res.setdefault(name, []).append(us)
    print(f"config{cid}: plain " + " ".join(f"{x:.2f}" for x in res["plain"]) + "   with terminal_obs " + " ".join(f"{x:.2f}" for x in res["term"]))
    env.close()
