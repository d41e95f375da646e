#!/usr/bin/env python3
"""Where does a launch spend its time OUTSIDE the life of one group?  (diagnostic build:
    python scripts/build_variant.py stamps -DDOCKAUV_STAMPS -DDOCKAUV_ROTATE_WAVES=0)
Every group records s_memrealtime (100 MHz, one clock for all XCDs) and s_memtime (shader ticks) at its first instruction
and behind its last store.  Printed: the launch as the groups see it (first start -> last end), the spread of the group
starts (dispatch ramp), the distribution of the group lives, and lives by position in the grid.
usage: DOCKAUV_LIB=.../libdockauv_stamps.so python scripts/span.py [--config 3] [--envs 65536]"""
import argparse, ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from gym_dockauv_amd.envs.batched import BatchedDocking3d
from gym_dockauv_amd import _capi

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=3)
ap.add_argument("--envs", type=int, default=0)
ap.add_argument("--threads", type=int, default=0)
ap.add_argument("--no-events", action="store_true", help="queue the launches without per-dispatch events")
ap.add_argument("--queued", type=int, default=1, help="launches queued back to back before the one that is read (1 = an isolated launch)")
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
G = min(16384, (N + 63) // 64)
for it in range(300):   # into the steady state of the episodes
    env.step_device(a[it % 8].data_ptr(), out.data_ptr(), stream=stream, packed=True)
torch.cuda.synchronize()
runs = []
seq = None
for it in range(40):
    # event-timed like bench.py: dockauv_time_steps on this very launch (the last of --queued back-to-back ones)
    # (--queued K: K launches queued by ONE C call, dockauv_time_steps, i.e. truly back to back on the stream -- launches
    # issued from Python arrive ~10 us apart and the GPU idles in between; `us` = their average event-timed duration)
    if args.no_events:   # the same K launches without per-dispatch events (dockauv_step_sequence: what bench.py's timed region queues)
        if seq is None:
            seq = env.make_step_sequence([a[q % 8].data_ptr() for q in range(args.queued)], [out.data_ptr()] * args.queued, packed=True)
        env.run_step_sequence(seq, stream=stream)
        us = float("nan")
    else:
        us = env.time_steps_device(a[it % 8].data_ptr(), out.data_ptr(), steps=args.queued, stream=stream, packed=True)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (6 * G))()
    rc = lib.dockauv_debug_read_span(buf, G)
    assert rc == 0, rc
    runs.append((us, np.frombuffer(buf, dtype=np.uint64).reshape(G, 6).astype(np.int64).copy()))
ev = np.array([r[0] for r in runs])
sp = np.stack([r[1] for r in runs])          # [launch, group, (rt, t) at entry / after the first kernarg + parameter words / end]
rt0, t0, rta, ta, rt1, t1 = (sp[:, :, k].astype(np.float64) for k in range(6))
arg_us = (rta - rt0) / 100.0
first = rt0.min(axis=1, keepdims=True)
start_us = (rt0 - first) / 100.0
end_us = (rt1 - first) / 100.0
life_ticks = t1 - t0
life_us = (rt1 - rt0) / 100.0
print(f"{wl['name']}  N={N}  groups={G}  (median over {len(runs)} launches, each the last of {args.queued} queued back to back; 100 MHz clock: +-0.01 us)")
print(f"  event-timed duration of these launches        : {np.nanmedian(ev):6.2f} us")
print(f"  first group start -> last group end           : {np.median(end_us.max(axis=1)):6.2f} us")
print(f"  first group start -> last group START (ramp)  : {np.median(start_us.max(axis=1)):6.2f} us   (p50 of the starts {np.median(np.median(start_us, axis=1)):.2f}, p90 {np.median(np.percentile(start_us, 90, axis=1)):.2f})")
print(f"  entry -> first kernarg / parameter words there: median {np.median(arg_us):.2f} us, p90 {np.median(np.percentile(arg_us, 90, axis=1)):.2f}, max {np.median(arg_us.max(axis=1)):.2f}")
print(f"  group life: median {np.median(life_us):.2f} us = {np.median(life_ticks):.0f} ticks  ({np.median(life_ticks) / np.median(life_us) / 1000:.3f} ticks/ns);"
      f"  p90 {np.median(np.percentile(life_us, 90, axis=1)):.2f}  p99 {np.median(np.percentile(life_us, 99, axis=1)):.2f}  max {np.median(life_us.max(axis=1)):.2f} us")
print(f"  first group END {np.median(end_us.min(axis=1)):.2f} us, median END {np.median(np.median(end_us, axis=1)):.2f}, p90 {np.median(np.percentile(end_us, 90, axis=1)):.2f}, last {np.median(end_us.max(axis=1)):.2f}")
# by position in the grid: groups are dealt to the XCDs round robin (group g -> XCD g % 8)
print("  by dispatch order (eighths of the grid): start / life / end, medians in us")
for k in range(8):
    sl = slice(k * G // 8, (k + 1) * G // 8)
    print(f"    groups {sl.start:5d}..{sl.stop - 1:5d}: start {np.median(start_us[:, sl]):5.2f}  life {np.median(life_us[:, sl]):5.2f}  end {np.median(end_us[:, sl]):5.2f}   max end {np.median(end_us[:, sl].max(axis=1)):5.2f}")
print("  by XCD (group % 8): start / life / end")
for x in range(8):
    print(f"    xcd {x}: start {np.median(start_us[:, x::8]):5.2f}  kernarg+params after {np.median(arg_us[:, x::8]):5.2f}  life {np.median(life_us[:, x::8]):5.2f}  max end {np.median(end_us[:, x::8].max(axis=1)):5.2f}")
env.close()
