"""PCIe-inclusive rate of the host-pointer entry point (dockauv_step_host: actions H2D, kernel, obs/reward/done D2H,
synchronous) -- the number DESIGN.md quotes next to the device-resident one.  Never bench.py's `value`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gym_dockauv_amd.envs.batched import BatchedDocking3d
for cfg_id, n in ((2, 4096), (2, 65536), (3, 65536)):
    wl = bench.workload(cfg_id, n)
    env = BatchedDocking3d(wl["cfg"], num_envs=n, scenario=wl["scenario"], precision="f32", reset_mode="device", rng="batched")
    env.reset()
    a = np.random.default_rng(0).uniform(-1, 1, (16, n, env.n_u)).astype(np.float32)
    for k in range(20):
        env.step(a[k % 16])
    t0 = time.perf_counter()
    K = 200
    for k in range(K):
        env.step(a[k % 16])
    dt = time.perf_counter() - t0
    import ctypes as C
    aa = np.ascontiguousarray(a[0], dtype=np.float32)
    io = env._io(aa, None, False)
    t1 = time.perf_counter()
    for k in range(K):
        env._lib.dockauv_step_host(env._handle, C.byref(io))
    dt_c = time.perf_counter() - t1
    print(f"  dockauv_step_host alone: {dt_c / K * 1e6:.1f} us/step = {n * K / dt_c:.3e} env-steps/s")
    print(f"host-pointer path, config {cfg_id}, N={n}: {dt / K * 1e6:.1f} us/step = {n * K / dt:.3e} env-steps/s (PCIe + Python inclusive)")
    env.close()
