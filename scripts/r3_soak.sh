#!/bin/bash
# soak: every BASELINE config's product kernel for ~2e5 steps back to back; rows finite, status word clean, auto-resets going
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
python - <<'PY' | tee gpurun_out/r3/soak.txt
import time, numpy as np, torch, bench
dev = torch.device("cuda", 0)
for cid in (2, 3, 4, 5):
    wl = bench.workload(cid, 0)
    env = bench.make_env(wl, 0, 0, 0)
    N, n, nu = wl["envs"], env.n_observations, env.n_u
    a = torch.rand((32, N, nu), device=dev) * 2 - 1
    out = torch.zeros((N, n + 2), device=dev)
    s = torch.cuda.current_stream().cuda_stream
    seq = env.make_step_sequence([a[i % 32].data_ptr() for i in range(2000)], [out.data_ptr()] * 2000, packed=True)
    t0 = time.perf_counter(); steps = 0; dones = 0
    while time.perf_counter() - t0 < float(__import__("os").environ.get("SOAK_SECONDS", "15")):
        env.run_step_sequence(seq, stream=s)
        torch.cuda.synchronize()
        steps += 2000
        assert bool(torch.isfinite(out).all().item()), f"config {cid}: non-finite rows after {steps} steps"
        dones += int((out[:, n + 1] > 0.5).sum().item())
    env.synchronize()          # raises DOCKAUV_E_KERNEL if a kernel ever set the status word
    ep = env.get_field(9)      # DOCKAUV_F_EPISODE
    print(f"config{cid}: {steps} steps x {N} envs = {steps * N:.3e} env-steps in {time.perf_counter() - t0:.1f} s, rows finite, status clean, "
          f"episodes per env: min {int(ep.min())} median {int(np.median(ep))} max {int(ep.max())}")
    env.close()
PY
