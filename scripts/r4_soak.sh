#!/bin/bash
# round 4 soak: every BASELINE config + configs 3 / 4 / 5 beyond residency (one wave per group; 4 / 5: capsule records in registers)
# for SOAK_SECONDS each, per launch AND as
# resident step sequences; rows finite, status word clean, auto-resets going -> gpurun_out/r4/soak.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
python - <<'PY' | tee gpurun_out/r4/soak.txt
import os, time, numpy as np, torch, bench
dev = torch.device("cuda", 0)
secs = float(os.environ.get("SOAK_SECONDS", "8"))
for cid, envs in ((2, 0), (3, 0), (4, 0), (5, 0), (3, 1048576), (4, 1048576), (5, 1048576), (4, 262144), (5, 524288)):
    for resident in (False, True):
        wl = bench.workload(cid, envs)
        env = bench.make_env(wl, 0, 0, 0)
        env.set_sequence_resident(resident)
        N, n, nu = wl["envs"], env.n_observations, env.n_u
        K = 2048 if N <= 65536 else 256
        a = torch.rand((16, N, nu), device=dev) * 2 - 1
        out = torch.zeros((N, n + 2), device=dev)
        s = torch.cuda.current_stream().cuda_stream
        seq = env.make_step_sequence([a[i % 16].data_ptr() for i in range(K)], [out.data_ptr()] * K, packed=True)
        t0 = time.perf_counter(); steps = 0
        while time.perf_counter() - t0 < secs:
            env.run_step_sequence(seq, stream=s)
            torch.cuda.synchronize()
            steps += K
            assert bool(torch.isfinite(out).all().item()), f"config {cid}: non-finite rows after {steps} steps"
        dt = time.perf_counter() - t0
        env.synchronize()          # raises DOCKAUV_E_KERNEL if a kernel ever set the status word
        ep = env.get_field(9)      # DOCKAUV_F_EPISODE
        print(f"config{cid} x {N} envs, {'resident sequences' if resident else 'one launch per step '}: {steps} steps = {steps * N:.3e} env-steps in {dt:.1f} s "
              f"({steps * N / dt:.3e} /s), rows finite, status clean, episodes per env: min {int(ep.min())} median {int(np.median(ep))} max {int(ep.max())}", flush=True)
        env.close()
PY
