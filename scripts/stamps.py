#!/usr/bin/env python3
"""In-kernel time line of the step kernel (diagnostic build: scripts/build_variant.py stamps -DDOCKAUV_STAMPS).
usage: DOCKAUV_LIB=.../libdockauv_stamps.so python scripts/stamps.py [--config 2] [--envs 4096]"""
import argparse, ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from gym_dockauv_amd.envs.batched import BatchedDocking3d
from gym_dockauv_amd import _capi

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
ap.add_argument("--envs", type=int, default=0)
ap.add_argument("--threads", type=int, default=0)
args = ap.parse_args()
wl = bench.workload(args.config, args.envs)
N = wl["envs"]
env = BatchedDocking3d(wl["cfg"], num_envs=N, scenario=wl["scenario"], device=0, precision="f32", reset_mode="device",
                       device_seed=1, rng="batched", vehicles=wl["vehicles"], threads_per_group=args.threads)
env.reset()
dev = torch.device("cuda", 0)
a = torch.rand((8, N, env.n_u), device=dev) * 2 - 1
out = torch.zeros((N, env.n_observations + 2), device=dev)
stream = torch.cuda.current_stream().cuda_stream
lib = _capi.load_library()
names = ["start", "loads issued", "loads landed + nu_c", "RK step", "ray stage", "nav + obs", "reward", "outputs", "reset + write-back", "obs tile store"]
acc = []
raw = []
for it in range(40):
    env.step_device(a[it % 8].data_ptr(), out.data_ptr(), stream=stream, packed=True)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (64 * 16))()
    rc = lib.dockauv_debug_read_stamps(buf)
    assert rc == 0, rc
    raw.append(bytes(buf))
    st = np.frombuffer(buf, dtype=np.uint64).reshape(64, 16)[:min(64, (N + 63) // 64), :10].astype(np.int64)
    if it >= 8:
        acc.append(st)
st = np.stack(acc)                       # [it, group, stamp]
d = np.diff(st, axis=2)                  # s_memtime ticks at 100 MHz? (shader clock on gfx950: see MICROARCH) -> report raw
print("s_memtime deltas per segment, median over groups and launches (ticks):")
for i in range(9):
    print(f"  {names[i]:>22s} -> {names[i+1]:<22s} {np.median(d[:, :, i]):9.0f}   (p10 {np.percentile(d[:, :, i], 10):7.0f}, p90 {np.percentile(d[:, :, i], 90):7.0f})")
full = np.stack([np.frombuffer(b, dtype=np.uint64).reshape(64, 16)[:min(64, (N + 63) // 64)].astype(np.int64) for b in raw[8:]])
if (full[:, :, 10] > 0).all():
    print("second wave of the group (ray stage only), relative to the group's start stamp:")
    for i, nm in ((14, "wave 0: part 1 done"), (15, "wave 0: records complete"), (10, "ray stage entered"), (11, "first cell done"), (12, "all cells done"), (13, "second barrier passed")):
        print(f"  {nm:>24s} at {np.median(full[:, :, i] - full[:, :, 0]):9.0f}")
print(f"  total {np.median(st[:, :, 9] - st[:, :, 0]):.0f} ticks; group start spread {np.median(st[:, :, 0].max(axis=1) - st[:, :, 0].min(axis=1)):.0f}")
env.close()
