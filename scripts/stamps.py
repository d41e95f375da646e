#!/usr/bin/env python3
"""In-kernel time line of the step kernel (diagnostic build:
    python scripts/build_variant.py stamps -DDOCKAUV_STAMPS -DDOCKAUV_ROTATE_WAVES=0).
usage: DOCKAUV_LIB=.../libdockauv_stamps.so python scripts/stamps.py [--config 2] [--envs 4096]
Stamps a launch did not write (roles a configuration does not have) are masked, not printed."""
import argparse, ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from gym_dockauv_amd.envs.batched import BatchedDocking3d
from gym_dockauv_amd import _capi

NS = 32   # stamps per group
ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
ap.add_argument("--envs", type=int, default=0)
ap.add_argument("--threads", type=int, default=0)
ap.add_argument("--shift", type=int, default=0, help="DOCKAUV_STAMP_SHIFT the library was built with")
ap.add_argument("--dense", action="store_true", help="ray-dense operating point (bench.py: place_ray_dense), no reset, holding inputs")
args = ap.parse_args()
wl = bench.workload(args.config, args.envs)
N = wl["envs"]
env = BatchedDocking3d(wl["cfg"], num_envs=N, scenario=wl["scenario"], device=0, precision="f32",
                       reset_mode="none" if args.dense else "device",
                       device_seed=1, rng="batched", vehicles=wl["vehicles"], threads_per_group=args.threads)
env._gen = np.random.default_rng(5)
env.reset()
dev = torch.device("cuda", 0)
a = torch.rand((8, N, env.n_u), device=dev) * 2 - 1
if args.dense:
    bench.place_ray_dense(env, np.random.default_rng(6))
    a.zero_()
    if wl["cfg"]["vehicle"] == "BlueROV2":
        a[:, :, 2] = 1.985 / 80.0
    wl["name"] += " -- ray-dense"
out = torch.zeros((N, env.n_observations + 2), device=dev)
stream = torch.cuda.current_stream().cuda_stream
lib = _capi.load_library()
G = min(64, ((N + 63) // 64 + (1 << args.shift) - 1) >> args.shift)
runs = []
for it in range(40):
    env.step_device(a[it % 8].data_ptr(), out.data_ptr(), stream=stream, packed=True)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (64 * NS))()
    rc = lib.dockauv_debug_read_stamps(buf)
    assert rc == 0, rc
    if it >= 8:
        runs.append(np.frombuffer(buf, dtype=np.uint64).reshape(64, NS)[:G].astype(np.int64).copy())
st = np.stack(runs)                                   # [launch, group, stamp]
rel = (st - st[:, :, :1]).astype(np.float64)          # relative to the group's start stamp
# a stamp this launch did not write is zero or left over from an earlier launch: before this launch's start
total = rel[:, :, 9:10].copy()
rel[rel < 0] = np.nan


def med(i):
    with np.errstate(all="ignore"):
        v = np.nanmedian(rel[:, :, i])
    return v


print(f"{wl['name']}  N={N}  (s_memtime ticks relative to the group's start; median over {G} groups x {len(runs)} launches)")
wave0 = [(6, "first kernarg word arrived"), (1, "loads issued"), (2, "loads landed, nu_c done"), (16, "inputs filtered (vehicle_step_)"), (17, "RHS 1"), (18, "RHS 2"),
         (19, "RHS 3"), (20, "RHS 4"), (21, "RHS 5"), (3, "RK step done"), (14, "pose trig / publish"), (15, "obstacle records complete"),
         (4, "tail of wave 0 entered"), (5, "tail of wave 0 done (nav obs, write-back)"), (7, "resetter entered"), (8, "reset + write-back issued"), (9, "obs tile stored = end")]
prev = 0.0
print("  wave 0 (integrating wave):")
for i, nm in wave0:
    v = med(i)
    if np.isnan(v):
        continue
    print(f"    {nm:<34s} at {v:8.0f}   (+{v - prev:6.0f})")
    prev = v
print("  second wave of the group:")
for i, nm in ((10, "ray stage entered / hand-over received"), (22, "active list built, free envs written"), (23, "ray passes done"), (11, "first cell done / reward done"), (12, "all cells done / tail waves done"), (13, "barrier after the ray stage passed")):
    v = med(i)
    if not np.isnan(v):
        print(f"    {nm:<42s} at {v:8.0f}")
arr = [med(24 + w) for w in range(8)]
if not all(np.isnan(a) for a in arr):
    print("  arrival of (physical) waves 0.. at the barrier behind the ray stage: " + "  ".join("-" if np.isnan(a) else f"{a:.0f}" for a in arr))
tt = total[:, :, 0]
print(f"  group life (start -> end): median {np.nanmedian(tt):.0f}  p90 {np.nanpercentile(tt, 90):.0f}  p99 {np.nanpercentile(tt, 99):.0f}  max {np.nanmax(tt):.0f}; "
      f"per launch, the slowest of the {G} sampled groups: median {np.nanmedian(np.nanmax(tt, axis=1)):.0f}")
# the slowest sampled group of each launch: where did it spend the extra time?
worst = np.nanargmax(tt, axis=1)
sel = rel[np.arange(rel.shape[0]), worst]          # [launch, stamp]
print("  slowest sampled group of each launch, medians: " + "  ".join(f"{nm.split()[0]}@{np.nanmedian(sel[:, i]):.0f}" for i, nm in
      ((2, "landed"), (3, "rk"), (14, "publish"), (15, "records"), (4, "raydone"), (5, "navobs"), (8, "wb"), (9, "end")) if not np.all(np.isnan(sel[:, i]))))
arr_w = [np.nanmedian(sel[:, 24 + w]) for w in range(8)]
if not all(np.isnan(a_) for a_ in arr_w):
    print("  slowest sampled group of each launch: arrival of (physical) waves 0.. at the barrier behind the ray stage: "
          + "  ".join("-" if np.isnan(a_) else f"{a_:.0f}" for a_ in arr_w)
          + f";  resetter entered@{np.nanmedian(sel[:, 7]):.0f}  second wave: ray stage entered@{np.nanmedian(sel[:, 10]):.0f} passes done@{np.nanmedian(sel[:, 23]):.0f}")
print(f"  total {np.nanmedian(total):.0f} ticks")
env.close()
