#!/bin/bash
# soak of the group shape dockauv_create picks for light fans beyond 524 288 envs (one wave per group): config 3 at 1 048 576 envs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
python - <<'PY' | tee gpurun_out/r3/soak_large.txt
import time, numpy as np, torch, bench
dev = torch.device("cuda", 0)
wl = bench.workload(3, 1048576)
env = bench.make_env(wl, 0, 0, 0)
N, n, nu = wl["envs"], env.n_observations, env.n_u
a = torch.rand((8, N, nu), device=dev) * 2 - 1
out = torch.zeros((N, n + 2), device=dev)
s = torch.cuda.current_stream().cuda_stream
seq = env.make_step_sequence([a[i % 8].data_ptr() for i in range(200)], [out.data_ptr()] * 200, packed=True)
t0 = time.perf_counter(); steps = 0
while time.perf_counter() - t0 < 20.0:
    env.run_step_sequence(seq, stream=s)
    torch.cuda.synchronize()
    steps += 200
    assert bool(torch.isfinite(out).all().item()), f"non-finite rows after {steps} steps"
env.synchronize()
ep = env.get_field(9)
print(f"config3: {steps} steps x {N} envs = {steps * N:.3e} env-steps in {time.perf_counter() - t0:.1f} s ({steps * N / (time.perf_counter() - t0):.3e} /s), rows finite, status clean, "
      f"episodes per env: min {int(ep.min())} median {int(np.median(ep))} max {int(ep.max())}")
env.close()
PY
