#!/usr/bin/env python3
"""
bench.py -- env-steps/s of the batched docking3d step() on N MI355X (BASELINE.json metric), one process per GPU.

  python bench.py --gpus 1 --steps K --warmup W                       (defaults finish in a few minutes)
  python bench.py --gpus N --steps K --warmup W                       (N > 1 without a launcher: starts its own N ranks)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

A "step" is one launch of the fused HIP step kernel over this rank's envs (weak scaling: the per-GPU env count is
fixed; at N = 1 the K launches of a region are queued by one dockauv_step_sequence call), followed -- for N > 1 --
by one gather of [obs | reward | done] over xGMI so that a single learner sees all observations.  Inputs (actions)
are resident in HBM before the timed region.  A timed region is EXACTLY K steps between barrier + synchronize on both
sides (max over ranks); regions are repeated until --min-seconds have been measured and the MEDIAN region is
reported (`reps`, `ms_per_step_first_region` and `ms_per_step_min` beside it), so that a short --steps is meaningful.
Rank 0 prints ONE JSON line.

Workloads (--config):
  2  BlueROV2, SimpleDocking3d, 4 096 envs/GPU, no obstacles (pure 6-DOF RKF45 + reward)
  3  BlueROV2, 16-beam fan vs 8 spheres, 65 536 envs/GPU          [default at N = 1: the largest single-GPU config]
  4  LAUV, ObstaclesDocking3d (5 capsules, 63 rays), t_step_size 0.02, 32 768 envs/GPU      [default at N > 1]
  5  BlueROV2/LAUV 50/50, ObstaclesCurrentDocking3d, current speed U(0,1), t_step_size 0.02, 65 536 envs/GPU
`roofline.kernel_us` = HIP events recorded on the launch stream around each timed region / K (the average launch duration,
launch boundary included; `--isolated-kernel-timing` adds the older per-dispatch events on launches issued one by one).
At N = 1 the line also carries `configs`: the same measurement (open-loop rate, kernel duration by HIP events,
roofline, closed-loop rate) of config 2 at 4 096 envs and of the per-GPU shards of configs 4 and 5.
"""
import argparse
import copy
import hashlib
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES = {2: 356, 3: 420, 4: 460, 5: 482}   # algorithmic bytes per env-step
HBM_PEAK_GBPS = 8000.0                           # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
N_SIMD = 1024                                    # 256 CUs x 4 SIMDs
CLOCK_MHZ = 2400.0                               # max shader clock (MI355X_MICROARCH.md)
VALU_CYCLES_PER_INST = 4.0                       # secondary pricing of a wave64 VALU instruction (rounds 1-3: x4, overstates two-fold)
VALU_CYCLES_MEASURED = 2.1                       # plain fp32 VALU instruction on a saturated SIMD, measured (scripts/micro/issue_rate.hip)


def workload(config_id: int, envs: int, layout: str = ""):
    from gym_dockauv_amd.config.env_config import BASE_CONFIG
    cfg = copy.deepcopy(BASE_CONFIG)
    fan16 = {"alpha": 30 * np.pi / 180, "beta": 30 * np.pi / 180, "ray_per_deg": 10 * np.pi / 180}
    if config_id == 2:
        return dict(id=2, cfg=cfg, scenario="SimpleDocking3d", envs=envs or 4096, vehicles=None, current_speed=None,
                    name="config2: BlueROV2 SimpleDocking3d, no sensors, h=0.1")
    if config_id == 3:
        cfg["radar"].update(fan16)
        return dict(id=3, cfg=cfg, scenario="SphereDocking3d", envs=envs or 65536, vehicles=None, current_speed=None,
                    name="config3: BlueROV2 + 16-beam fan vs 8 spheres, h=0.1")
    if config_id == 4:
        cfg["vehicle"] = "LAUV"
        cfg["t_step_size"] = 0.02
        return dict(id=4, cfg=cfg, scenario="ObstaclesDocking3d", envs=envs or 32768, vehicles=None, current_speed=None,
                    name="config4: LAUV ObstaclesDocking3d (5 capsules, 63 rays), h=0.02")
    if config_id == 5:
        cfg["t_step_size"] = 0.02
        n = envs or 65536
        if layout == "vehicle_sorted" or (not layout and os.environ.get("DOCKAUV_CONFIG5_SORTED") == "1"):
            # vehicle-sorted layout: first half BlueROV2, second half LAUV -> every wave but one is homogeneous
            return dict(id=5, cfg=cfg, scenario="ObstaclesCurrentDocking3d", envs=n, current_speed="uniform01",
                        vehicles=["BlueROV2"] * (n // 2) + ["LAUV"] * (n - n // 2),
                        name="config5: BlueROV2/LAUV vehicle-sorted, ObstaclesCurrentDocking3d, V_c~U(0,1), h=0.02")
        return dict(id=5, cfg=cfg, scenario="ObstaclesCurrentDocking3d", envs=n, current_speed="uniform01",
                    vehicles=["BlueROV2" if i % 2 == 0 else "LAUV" for i in range(n)],
                    name="config5: BlueROV2/LAUV interleaved, ObstaclesCurrentDocking3d, V_c~U(0,1), h=0.02")
    raise SystemExit(f"unknown --config {config_id}")


def kernel_source_sha() -> str:
    """Identifies the kernel version a profile under profiles/ belongs to: a hash of the device sources without their
    comments (// and /* */, string literals respected) and blank lines (a comment edit does not make the committed
    counters stale)."""
    import re
    h = hashlib.sha256()
    d = os.path.join(ROOT, "gym_dockauv_amd", "csrc")
    pat = re.compile(r'//[^\n]*|/\*.*?\*/|"(?:\\.|[^"\\])*"|\'(?:\\.|[^\'\\])*\'', re.S)
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".inc", ".h")):
            text = open(os.path.join(d, f), encoding="utf-8", errors="replace").read()
            text = pat.sub(lambda m: m.group(0) if m.group(0)[0] in "\"'" else " ", text)
            for line in text.splitlines():
                code = line.rstrip()
                if code.strip():
                    h.update(code.encode() + b"\n")
    return h.hexdigest()[:12]


def pmc_entry(config_id: int, envs: int, variant: str = ""):
    """Counters of the step kernel for this workload from the committed rocprofv3 --pmc passes (scripts/profile_round.sh
    -> profiles/pmc_counters.json): they are NOT measured in this run, so the entry carries its source."""
    path = os.path.join(ROOT, "profiles", "pmc_counters.json")
    try:
        e = json.load(open(path)).get(f"config{config_id}{'_' + variant if variant else ''}_envs{envs}")
    except Exception:
        return None
    return e


# ------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(wl, seconds: float):
    """The NumPy oracle (a port of the reference's per-env NumPy path) timed on this host, one core, on a bounded
    sample of the same workload (same scenario, vehicle, fan, step size).  Reported next to the GPU number; it is
    NOT on the product path."""
    from oracle import dockauv_oracle as orc
    scenario = wl["scenario"]
    if scenario not in orc.SCENARIOS:
        raise SystemExit(f"oracle has no scenario {scenario}")
    vehs = sorted(set(wl["vehicles"])) if wl["vehicles"] else [wl["cfg"]["vehicle"]]
    envs = []
    for veh in vehs:
        e = orc.OracleEnv(scenario, {"vehicle": veh, "t_step_size": wl["cfg"]["t_step_size"],
                                     "radar": {k: wl["cfg"]["radar"][k] for k in ("alpha", "beta", "ray_per_deg", "max_dist")}})
        e.reset(seed=0)
        envs.append(e)
    rs = np.random.RandomState(1234)

    def some_steps(k):
        for e in envs:
            for _ in range(k):
                _, _, d, _ = e.step(rs.uniform(-1, 1, e.model.n_u))
                if d:
                    e.reset()
    some_steps(50)
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        some_steps(50)
        n += 50 * len(envs)
    dt = time.perf_counter() - t0
    n_sph = len(envs[0].sphere_radii)
    return {"value": n / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"oracle/dockauv_oracle.py (NumPy, float64, 1 core), {n} steps of {scenario}/{'+'.join(vehs)} "
                      f"({envs[0].fan.n_rays} rays, {len(envs[0].capsules)} capsules, {n_sph} spheres) in {dt:.1f} s, "
                      f"uniform random actions, reset on done",
            "reference_numpy_measured_in_survey": "~270 env-steps/s/core (SimpleDocking3d), ~250 (ObstaclesDocking3d): BASELINE.md section 2"}


def cpu_baseline_all_cores(config_id: int, seconds: float):
    """The same oracle loop in one child process per CPU of this host (independent envs, as SURVEY.md section 8d asks);
    children never touch the GPU.  Returns (aggregate env-steps/s, processes)."""
    import subprocess
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    procs = max(1, min(avail, 16))   # a one-GPU box's CPU share is 16 cores
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", "--config", str(config_id),
           "--cpu-seconds", str(seconds)]
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="")
    children = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env) for _ in range(procs)]
    total = 0.0
    for c in children:
        out, _ = c.communicate(timeout=seconds * 6 + 60)
        try:
            total += float(out.strip().splitlines()[-1])
        except (ValueError, IndexError):
            pass
    return total, procs


# ------------------------------------------------------------------------------------------ GPU helpers
def make_env(wl, local_rank: int, rank: int, threads: int):
    from gym_dockauv_amd.envs.batched import BatchedDocking3d
    from gym_dockauv_amd import _capi
    env = BatchedDocking3d(wl["cfg"], num_envs=wl["envs"], scenario=wl["scenario"], device=local_rank, precision="f32",
                           reset_mode=os.environ.get("DOCKAUV_BENCH_RESET_MODE", "device"),   # (diagnostic override: "none")
                           device_seed=0x5EED0000 + rank, rng="batched", vehicles=wl["vehicles"],
                           threads_per_group=threads)
    env._gen = np.random.default_rng(1000 + rank)
    env.reset()
    # `value` and every per-launch figure: one launch per step (what a policy in the loop gets); the resident fast path of
    # dockauv_step_sequence is measured on its own (`sequence_resident`)
    env.set_sequence_resident(False)
    if wl["current_speed"] == "uniform01":
        # per-env current speed U(0, 1) (the scenario itself fixes 0.5 m/s, docking3d.py:985);
        # V_min = V_max = V_c as in SimpleCurrentDocking3d (docking3d.py:844-848).  Episodes generated in-kernel after
        # a reset use the scenario's own 0.5 m/s.
        cur = env.get_field(_capi.F_CURRENT)
        v = np.random.default_rng(77 + rank).random(wl["envs"])
        cur[:, 0] = cur[:, 1] = cur[:, 2] = v
        env.set_field(_capi.F_CURRENT, cur)
    return env


# What limits each BASELINE workload at its batch size: the figure reported is always the
# algorithmic-bytes fraction of the HBM peak (the contract's `achieved` / `peak`), `bound` says what the launch is really
# waiting for.  "hbm": streaming bound.  "valu": above the ridge (~48 FLOP/B for the 63-ray fan), the ray stage's
# arithmetic at 4 waves per SIMD.  "latency": every group of the launch is resident at once and the launch lasts as long
# as ONE group's dependent instruction stream (a lone integrating wave per SIMD).  "launch": fewer waves than SIMDs; the
# event-timed duration sits on the floor of an empty dispatch (profiles/r2/launch_floor.txt: 3.96 us).
def bound_of(config_id: int, envs: int, dense: bool = False) -> str:
    if envs <= 8192:
        return "launch"
    if envs > 262144:
        return "hbm" if config_id == 2 else "latency"     # (ray kernels: LDS-limited residency x group latency)
    if dense and config_id in (4, 5):
        return "valu"
    return "latency"


def roofline_of(config_id: int, envs: int, kernel_us: float, n_timed: int, ksha: str, dense: bool = False, variant: str = ""):
    bytes_per_launch = ALGO_BYTES[config_id] * envs
    achieved = bytes_per_launch / (kernel_us * 1e-6) / 1e9
    r = {"bound": "hbm", "limited_by": bound_of(config_id, envs, dense),
         "limited_by_source": "design-time classification by config and batch size (DESIGN.md section 5): not derived from this run",
         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
         "traffic": None, "kernel": "dockauv::step_kernel", "kernel_us": kernel_us,
         "algorithmic_bytes_per_env_step": ALGO_BYTES[config_id], "envs_per_launch": envs,
         "timing": (KERNEL_TIMING + f"; {n_timed} launches") if isinstance(n_timed, int) else str(n_timed)}
    # what the launch touches (state, obstacles, the action ring, rows): below the 256 MiB Infinity Cache the "HBM" fraction is
    # measured on cache-resident data (FETCH_SIZE counts MALL hits as well); the sweep's 1 048 576-env point is the one that
    # streams from HBM proper
    ws = bytes_per_launch * 1.3
    r["working_set_note"] = (f"~{ws / 2**20:.0f} MiB touched per launch: " +
                             ("fits the 256 MiB Infinity Cache -- the HBM-resident operating points are the sweep's (262 144 / 1 048 576 envs)"
                              if ws < 200 * 2**20 else "beyond the 256 MiB Infinity Cache: streams from HBM"))
    e = pmc_entry(config_id, envs, "dense" if dense else variant)
    if e:
        stale = e.get("kernel_sha") != ksha
        r["traffic"] = e.get("traffic_bytes")
        r["traffic_source"] = (f"{e.get('source', 'profiles/pmc_counters.json')} (rocprofv3 --pmc passes of kernel "
                               f"{e.get('kernel_sha')}, not this run" + ("; THIS library is a later kernel version" if stale else "") + ")")
        if e.get("traffic_bytes"):
            r["traffic_over_algorithmic"] = e["traffic_bytes"] / bytes_per_launch
        if e.get("sq_insts_valu"):
            if stale:
                r["valu_frac_note"] = "SQ_INSTS_VALU was counted on an earlier kernel version than this library: indicative only"
            cyc = kernel_us * CLOCK_MHZ
            # share of the SIMDs' cycles spent issuing VALU instructions, priced at what a saturated SIMD was MEASURED to need
            # for a plain fp32 wave64 instruction (2.1 cycles: 157 TFLOP/s / 1 024 SIMDs / 2.4 GHz = 64 flop per clock and SIMD;
            # profiles/r2/issue_rate.txt; packed, DPP and transcendental instructions cost 2-4 x that) -- a lower bound of the
            # busy share.  The x 4 pricing of rounds 1-3 (one instruction per 4 cycles) overstated it two-fold and stays as a
            # secondary figure.
            r["valu_frac"] = e["sq_insts_valu"] * VALU_CYCLES_MEASURED / (N_SIMD * cyc)
            r["valu_frac_at_4_cycles_per_inst"] = e["sq_insts_valu"] * VALU_CYCLES_PER_INST / (N_SIMD * cyc)
            r["valu_frac_inputs"] = {"SQ_INSTS_VALU_per_launch": e["sq_insts_valu"], "cycles_per_inst": VALU_CYCLES_MEASURED,
                                     "cycles_per_inst_secondary": VALU_CYCLES_PER_INST, "simds": N_SIMD, "kernel_cycles_at_2400MHz": cyc}
    return r


def timed_regions(run, steps, warmup_steps, warm_i0, min_seconds, max_reps, sync, barrier, reduce_max, event_ms=None, torch=None):
    """Warm up, then time regions of EXACTLY `steps` steps (barrier + synchronize on both sides, max over ranks) until
    `min_seconds` have been measured.  Returns the list of region durations (host clock).  event_ms (a list): each
    region is also bracketed by two HIP events recorded on the launch stream (torch's current stream: the one the steps are
    queued on); their elapsed times in ms are appended to it."""
    if warmup_steps > 0:
        run(warmup_steps, warm_i0)
    times = []
    while True:
        sync(); barrier(); sync()
        if event_ms is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        t0 = time.perf_counter()
        run(steps, warm_i0 + warmup_steps)
        if event_ms is not None:
            ev1.record()
        sync(); barrier(); sync()
        times.append(reduce_max(time.perf_counter() - t0))
        if event_ms is not None:
            event_ms.append(ev0.elapsed_time(ev1))
        if sum(times) >= min_seconds or len(times) >= max_reps:
            return times


KERNEL_TIMING = ("HIP events recorded on the launch stream around each timed region of K launches queued back to back by one "
                 "dockauv_step_sequence call: median region time / K = the average launch duration, launch boundary included "
                 "(an upper bound of the kernel's own duration; rocprofv3's queued dispatches report the same quantity)")


def resident_sequence_rate(env, run, args, torch, config_id, N, per_launch_us):
    """The same timed regions with dockauv_step_sequence's RESIDENT fast path: launches of up to 64 steps in which every
    64-env group walks its envs through all of them -- no launch boundary between steps (include/dockauv.h).  Open-loop
    sequences only; reported beside the per-launch `value`, never instead of it."""
    env.set_sequence_resident(True)
    ev_ms = []
    times = timed_regions(run, args.steps, min(args.warmup, 128), 0, args.min_seconds / 2, args.max_reps, torch.cuda.synchronize,
                          lambda: None, lambda x: x, event_ms=ev_ms, torch=torch)
    env.set_sequence_resident(False)
    med = statistics.median(times)
    us = statistics.median(ev_ms) / args.steps * 1e3
    gbps = ALGO_BYTES[config_id] * N / (us * 1e-6) / 1e9
    return {"value": N * args.steps / med, "unit": "env-steps/s", "ms_per_step": med / args.steps * 1e3, "us_per_step_events": us,
            "reps": len(times), "steps_per_launch": 64, "achieved_GBps_algorithmic": gbps, "frac_of_8TBps": gbps / HBM_PEAK_GBPS,
            "vs_per_launch": us / per_launch_us if per_launch_us else None,
            "what": "dockauv_step_sequence with its resident fast path: K steps as ceil(K / 64) launches, each group walking its 64 "
                    "envs through the launch's steps (state through L2 between steps, rows and actions as in K launches: same bytes)"}


def isolated_kernel_us(env, actions, out, ring, stream, torch, launches):
    """The older measure (--isolated-kernel-timing): start / stop events attached to single dispatches launched one by one
    with a host synchronisation in between (dockauv_time_steps).  It carries what an isolated launch pays on top: XCDs
    that start up to ~2 us late after an idle gap, and with per-dispatch events XCD 0 lagging 0.6-0.8 us."""
    us = 0.0
    for i in range(launches):
        us += env.time_steps_device(actions[i % ring].data_ptr(), out.data_ptr(), steps=1, stream=stream, packed=True)
    torch.cuda.synchronize()
    return us / launches


def closed_loop_rate(env, torch, dev, N, n_obs, n_u, steps):
    """Closed loop: the actions of step t + 1 are computed ON THE DEVICE from the observations of step t (a linear
    policy + tanh: torch kernels between the step kernels).  Two forms: issued step by step from Python, and the same
    K steps captured once in a HIP graph and replayed (no host work per step)."""
    out = torch.zeros((N, n_obs + 2), device=dev, dtype=torch.float32)
    W = torch.randn((n_obs, n_u), device=dev, generator=torch.Generator(device=dev).manual_seed(5)) * 0.3
    res = {"policy": "tanh(obs @ W), torch on the same stream"}

    def loop(k, stream):
        for _ in range(k):
            a = torch.tanh(out[:, :n_obs] @ W)
            env.step_device(a.data_ptr(), out.data_ptr(), stream=stream, packed=True)
        return a

    s0 = torch.cuda.current_stream().cuda_stream
    loop(20, s0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(steps, s0)
    torch.cuda.synchronize()
    res["python_issued_us_per_step"] = (time.perf_counter() - t0) / steps * 1e6
    try:
        K = 50
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            keep = loop(K, torch.cuda.current_stream().cuda_stream)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        reps = max(1, steps // K)
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
        res["hip_graph_us_per_step"] = (time.perf_counter() - t0) / (reps * K) * 1e6
        del keep, g
    except Exception as ex:   # graph capture is an optimisation of the caller, not of the path
        res["hip_graph_us_per_step"] = None
        res["hip_graph_note"] = f"capture failed: {type(ex).__name__}: {ex}"[:200]
        torch.cuda.synchronize()
    best = min(v for v in (res["python_issued_us_per_step"], res.get("hip_graph_us_per_step")) if v)
    res["us_per_step"] = best
    res["env_steps_per_s"] = N / (best * 1e-6)
    return res


def place_ray_dense(env, rng):
    """Every env within sensor range of one of its obstacles and facing it (host set_field): 3-5 m from the surface of its
    first sphere, or from the axis-parallel surface of its first capsule, at the obstacle's depth, heading towards it
    (+- 0.1 rad), at rest.  Where a docking policy lives: the goal sits ON the centre capsule's safety surface
    (envs/docking3d.py:868-876)."""
    from gym_dockauv_amd import _capi
    N = env.num_envs
    ang = rng.uniform(-np.pi, np.pi, N)
    if env.max_spheres:
        sph = env.get_field(_capi.F_SPHERES).reshape(N, -1, 4)
        centre, rad = sph[:, 0, 0:3], sph[:, 0, 3]
    else:
        caps = env.get_field(_capi.F_CAPSULES).reshape(N, -1, 7)
        centre = 0.5 * (caps[:, 0, 0:3] + caps[:, 0, 3:6])
        centre[:, 2] += rng.uniform(-1.0, 1.0, N)
        rad = caps[:, 0, 6]
    d = rad + rng.uniform(3.0, 5.0, N)
    state = np.zeros((N, 12))
    state[:, 0] = centre[:, 0] - d * np.cos(ang)
    state[:, 1] = centre[:, 1] - d * np.sin(ang)
    state[:, 2] = centre[:, 2]
    psi = ang + rng.uniform(-0.1, 0.1, N)
    state[:, 5] = (psi + np.pi) % (2 * np.pi) - np.pi
    env.set_field(_capi.F_STATE, state)
    env.set_field(_capi.F_U, np.zeros((N, 8)))
    env.set_field(_capi.F_TSTEPS, np.zeros((N, 1)))


def measure_ray_dense(config_id, args, torch, dev, local_rank, rank, ksha):
    """The ray stage where it is busy: under uniform random actions 84-99 % of the envs have no obstacle
    within reach of the fan and take no ray pass at all (profiles/r2/active_fraction.txt).  Here every env sits 3-5 m in
    front of an obstacle and holds its position (BlueROV2: the heave input that cancels its 1.985 N of buoyancy, LAUV:
    zero inputs; no reset; vehicles are put back before every timed region), so that >= 90 % of the fans have a hit."""
    from gym_dockauv_amd.envs.batched import BatchedDocking3d
    wl = workload(config_id, 0)
    N = wl["envs"]
    env = BatchedDocking3d(wl["cfg"], num_envs=N, scenario=wl["scenario"], device=local_rank, precision="f32",
                           reset_mode="none", rng="batched", vehicles=wl["vehicles"], threads_per_group=args.threads)
    env._gen = np.random.default_rng(2000 + rank)
    env.reset()
    env.set_sequence_resident(False)   # (per-launch figures, as everywhere in this line but `sequence_resident`)
    rng = np.random.default_rng(3000 + rank)
    n_obs, n_u = env.n_observations, env.n_u
    hold = torch.zeros((N, n_u), device=dev, dtype=torch.float32)
    if wl["cfg"]["vehicle"] == "BlueROV2":
        hold[:, 2] = 1.985 / 80.0     # W - B = -1.985 N (Q14); heave gain 4 x 20 N per unit input (BlueROV2.py:34-51)
    out = torch.zeros((N, n_obs + 2), device=dev, dtype=torch.float32)
    stream = torch.cuda.current_stream().cuda_stream
    K = min(args.steps, 250)
    seq = env.make_step_sequence([hold.data_ptr()] * K, [out.data_ptr()] * K, packed=True)

    hold_np = hold.cpu().numpy().astype(np.float64)

    def active_fraction():
        """share of the envs with a ray that hits within range, from the clamped ray distances of one more step on the
        full-output path (a cell of the observation is the block MAX of 2 x 2 rays, sensor.py:131-137: it reads < 1 only
        when all four hit -- reported beside it)"""
        _, _, _, _ = env.step(hold_np, extras=True)
        hit = float((env.intersec_dist < env.radar.max_dist).any(axis=1).mean())
        cells = float((out[:, 16:n_obs] < 1.0).any(dim=1).float().mean().item())
        return hit, cells

    place_ray_dense(env, rng)
    env.run_step_sequence(seq, stream=stream)
    torch.cuda.synchronize()
    times, act_end, ev_ms = [], [], []
    while sum(times) < args.min_seconds / 4 and len(times) < args.max_reps:
        place_ray_dense(env, rng)
        env.step_device(hold.data_ptr(), out.data_ptr(), stream=stream, packed=True)
        torch.cuda.synchronize()
        act0 = active_fraction()[0]
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        t0 = time.perf_counter()
        env.run_step_sequence(seq, stream=stream)
        ev1.record()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        ev_ms.append(ev0.elapsed_time(ev1))
        act_end.append(active_fraction()[0])
    med = statistics.median(times)
    kernel_us, n_k = statistics.median(ev_ms) / K * 1e3, len(ev_ms) * K
    res = {"workload": wl["name"] + " -- ray-dense operating point", "envs": N, "value": N * K / med, "unit": "env-steps/s",
           "ms_per_step": med / K * 1e3, "reps": len(times), "steps_per_region": K, "kernel_us": kernel_us,
           "active_fraction": {"what": "share of the envs with at least one ray that hits within range (clamped distance < max_dist)",
                               "region_start": act0, "region_end_min": min(act_end), "after_kernel_timing": active_fraction()[0],
                               "envs_with_a_cell_below_1": active_fraction()[1]},
           "roofline": roofline_of(config_id, N, kernel_us, n_k, ksha, dense=True), "obs_finite": bool(torch.isfinite(out).all().item())}
    env.close()
    return res


# configs 4 / 5 beyond residency: BASELINE's total env count of the config (what its eight GPUs share) and 1 048 576
BEYOND_RESIDENCY = {4: (262144, 1048576), 5: (524288, 1048576)}


def sweep_point(wl, config_id, n_big, n_base, args, torch, dev, local_rank, stream, gen):
    """The workload's kernel at a batch size beyond residency (launches of several rounds of groups; the state no longer fits
    the caches: the HBM-relevant roofline points), per launch and as resident sequences: regions of 128 steps."""
    from gym_dockauv_amd.envs.batched import BatchedDocking3d
    e2 = BatchedDocking3d(wl["cfg"], num_envs=n_big, scenario=wl["scenario"], device=local_rank, precision="f32",
                          reset_mode="device", device_seed=0xABC, rng="batched",
                          vehicles=(wl["vehicles"] * (n_big // n_base + 1))[:n_big] if wl["vehicles"] else None)
    e2._gen = np.random.default_rng(7)
    e2.reset()
    e2.set_sequence_resident(False)
    n_obs, n_u = e2.n_observations, e2.n_u
    a2 = torch.rand((4, n_big, n_u), device=dev, generator=gen, dtype=torch.float32) * 2 - 1
    o2 = torch.zeros((n_big, n_obs + 2), device=dev, dtype=torch.float32)
    KS = 128   # (regions of 20 launches, rounds 1-3, carried the first launches' cold caches: 121 us against 108)
    seq2 = e2.make_step_sequence([a2[r % 4].data_ptr() for r in range(KS)], [o2.data_ptr()] * KS, packed=True)

    def region_us(reps):
        e2.run_step_sequence(seq2, stream=stream)
        torch.cuda.synchronize()
        ev = []
        for r in range(reps):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            e2.run_step_sequence(seq2, stream=stream)
            ev1.record()
            torch.cuda.synchronize()
            ev.append(ev0.elapsed_time(ev1) / KS * 1e3)
        return statistics.median(ev)
    us = region_us(5)
    e2.set_sequence_resident(True)
    us_res = None if args.no_resident else region_us(5)
    threads = getattr(e2, "threads_in_use", None)
    gbps = ALGO_BYTES[config_id] * n_big / (us * 1e-6) / 1e9
    res = {"envs": n_big, "kernel_us": us, "env_steps_per_s_kernel": n_big / (us * 1e-6),
           "achieved_GBps": gbps, "frac_of_8TBps": gbps / HBM_PEAK_GBPS,
           "sequence_resident_us_per_step": us_res,
           "sequence_resident_frac_of_8TBps": (ALGO_BYTES[config_id] * n_big / (us_res * 1e-6) / 1e9 / HBM_PEAK_GBPS) if us_res else None}
    if threads:
        res["threads_per_group"] = threads
    e2.close()
    del a2, o2
    return res


def measure_single(config_id, envs, args, torch, dev, local_rank, rank, ksha, kernel_launches=512, layout=""):
    """One workload on this GPU alone (no collective): open-loop regions, kernel duration by events, closed loop."""
    wl = workload(config_id, envs, layout)
    N = wl["envs"]
    env = make_env(wl, local_rank, rank, args.threads)
    n_obs, n_u = env.n_observations, env.n_u
    RING = 64 if N <= 65536 else 16
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    actions = torch.rand((RING, N, n_u), device=dev, generator=gen, dtype=torch.float32) * 2 - 1
    out = torch.zeros((N, n_obs + 2), device=dev, dtype=torch.float32)
    stream = torch.cuda.current_stream().cuda_stream
    cache = {}

    def run(n, i0=0):
        key = (n, i0 % RING)
        if key not in cache:
            cache[key] = env.make_step_sequence([actions[i % RING].data_ptr() for i in range(i0, i0 + n)], [out.data_ptr()] * n, packed=True)
        env.run_step_sequence(cache[key], stream=stream)

    ev_ms = []
    times = timed_regions(run, args.steps, args.warmup, 0, args.min_seconds / 2, args.max_reps, torch.cuda.synchronize,
                          lambda: None, lambda x: x, event_ms=ev_ms, torch=torch)
    med = statistics.median(times)
    kernel_us = statistics.median(ev_ms) / args.steps * 1e3
    finite = bool(torch.isfinite(out).all().item())
    iso = isolated_kernel_us(env, actions, out, RING, stream, torch, kernel_launches) if (args.isolated_kernel_timing or args.steps < 100) else None
    if args.steps < 100:
        kernel_us = iso
    cl = None if args.no_closed_loop else closed_loop_rate(env, torch, dev, N, n_obs, n_u, 400)
    resident = None if args.no_resident else resident_sequence_rate(env, run, args, torch, config_id, N, kernel_us)
    res = {"workload": wl["name"], "envs": N, "value": N * args.steps / med, "unit": "env-steps/s",
           "ms_per_step": med / args.steps * 1e3, "reps": len(times), "kernel_us": kernel_us,
           "roofline": roofline_of(config_id, N, kernel_us, len(ev_ms) * args.steps, ksha, variant="sorted" if layout == "vehicle_sorted" else ""),
           "closed_loop": cl, "sequence_resident": resident, "obs_finite": finite}
    if iso is not None:
        res["kernel_us_isolated_launches"] = iso
    env.close()
    del actions, out
    if config_id in BEYOND_RESIDENCY and not layout and not envs and not args.no_sweep:
        # the 8-GPU workloads as ONE GPU would run them whole, and at 1 048 576 envs (one wave per group there: dockauv_create)
        res["beyond_residency"] = [sweep_point(wl, config_id, n_big, N, args, torch, dev, local_rank, stream, gen)
                                   for n_big in BEYOND_RESIDENCY[config_id]]
    return res


def split_json_line(out: str):
    """(the LAST line of `out` that is a JSON object, every other line) -- (None, []) if there is none."""
    lines = out.splitlines()
    js = [k for k, ln in enumerate(lines) if ln.lstrip().startswith("{") and ln.rstrip().endswith("}")]
    if not js:
        return None, []
    return lines[js[-1]].strip(), [ln for k, ln in enumerate(lines) if k != js[-1]]


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` started without a launcher: become one.  N child rank processes of this very command line
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1) are started BEFORE this process imports torch
    or makes any HIP call; rank 0's stdout -- the ONE JSON line -- is relayed, the other ranks' stdout goes to stderr.
    Returns 0 only if every rank did; a rank that fails takes the others down (they would wait at the rendezvous)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # (dmabuf IPC: what RCCL needs on this driver)
    base.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    children = []
    share = os.environ.get("DOCKAUV_RANKS_SHARE_GPU") == "1"   # rehearsal on a one-GPU box (with DOCKAUV_DIST_BACKEND=gloo)
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK="0" if share else str(r))
        children.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr))
    import threading
    chunks = []
    drain = threading.Thread(target=lambda: chunks.append(children[0].stdout.read()), daemon=True)   # (a full pipe must not block rank 0)
    drain.start()
    rc = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            code = children[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} exited with {code}: stopping the other ranks", file=sys.stderr, flush=True)
                for q in pending:
                    children[q].terminate()   # (exactly the processes started above)
        if pending:
            time.sleep(0.05)
    drain.join(timeout=30)
    out = b"".join(chunks).decode(errors="replace")
    if rc == 0:
        # ONE JSON line on stdout: whatever else rank 0's stdout carried (a collective library's banner, e.g. "[Gloo] Rank 0
        # is connected to ...") goes to stderr
        line, others = split_json_line(out)
        if others:
            sys.stderr.write("\n".join(others) + "\n")
            sys.stderr.flush()
        sys.stdout.write(line + "\n" if line is not None else out)
        sys.stdout.flush()
    else:
        sys.stderr.write(out)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--config", type=int, default=0, help="0 = config 3 at N = 1, config 4 (32 768 envs per GPU) at N > 1")
    ap.add_argument("--envs", type=int, default=0, help="envs per GPU (0 = the config's size)")
    ap.add_argument("--min-seconds", type=float, default=0.2, help="timed regions are repeated until this much has been measured")
    ap.add_argument("--max-reps", type=int, default=4000)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the sub-results of the other BASELINE configs")
    ap.add_argument("--cpu-worker", action="store_true", help=argparse.SUPPRESS)   # child of cpu_baseline_all_cores
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--no-overlap", action="store_true", help="do not overlap the all-gather with the next kernel")
    ap.add_argument("--gather", choices=["auto", "rccl", "p2p"], default="rccl",
                    help="N > 1: transport of the per-step gather.  rccl (default) = one all_gather_into_tensor per step, "
                         "as BASELINE's north_star names it; auto = peer-to-peer push (dockauv_p2p_*) if it maps and "
                         "reproduces the RCCL all-gather bit for bit during warm-up, else RCCL (the links bound the step "
                         "either way, DESIGN.md section 7; the push has only run with ranks sharing one GPU)")
    ap.add_argument("--gather-dtype", choices=["f32", "bf16"], default="f32",
                    help="N > 1: what crosses xGMI per env and step.  f32 (default): the packed float32 rows, bit for bit; "
                         "bf16: observation columns as bfloat16 (half the bytes; reward / done stay float32).  The f32 line "
                         "also carries a `bf16_gather` sub-measurement of the same regions")
    ap.add_argument("--no-closed-loop", action="store_true",
                    help="skip the closed-loop sub-measurement (its launches are issued from Python one by one: profiling runs "
                         "use this so that rocprofv3's per-kernel average covers the queued launches of the timed regions only)")
    ap.add_argument("--no-resident", action="store_true", help="skip the sequence_resident sub-measurement")
    ap.add_argument("--isolated-kernel-timing", action="store_true",
                    help="also report kernel_us_isolated_launches: per-dispatch events on launches issued one by one")
    ap.add_argument("--only-ray-dense", type=int, default=0, metavar="CONFIG",
                    help="profiling aid: run only the ray-dense measurement of config 3 or 4 and print its JSON")
    ap.add_argument("--layout", default="", help="config 5: '' / interleaved (default) or vehicle_sorted")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--sweep", type=int, nargs="*", default=[262144, 1048576])
    args = ap.parse_args()
    if args.cpu_worker:      # oracle loop only: no torch, no GPU
        print(cpu_baseline(workload(args.config or 3, 2), args.cpu_seconds)["value"])
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process becomes the launcher (nothing has touched the GPU yet)
        sys.exit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE = {world}: start it as `python bench.py --gpus N` (it launches its own "
                         "ranks) or as `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    if world > 1 and os.environ.get("DOCKAUV_DIST_BACKEND", "nccl") == "gloo":
        # rehearsal transport (several ranks on one GPU, or none: tests/test_bench_contract.py): the rendezvous needs no device,
        # so it is made -- and proven by a collective -- before the first thing that does
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        t = torch.tensor([rank + 1], dtype=torch.int64)
        dist.all_reduce(t)
        assert int(t.item()) == world * (world + 1) // 2
        print(f"[rank {rank}] rendezvous of {world} ranks complete (gloo)", file=sys.stderr, flush=True)
    if not torch.cuda.is_available():
        raise SystemExit(f"[rank {rank}] bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    config_id = args.config or (3 if world == 1 else 4)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("DOCKAUV_FORCE_DIST") == "1"   # the latter: 1-rank rehearsal of the RCCL path
    saved_stdout = None
    backend = None
    if use_dist:
        # RCCL prints a version banner on stdout when the communicator is created: keep stdout for the ONE JSON line
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if os.environ.get("DOCKAUV_DIST_BACKEND", "nccl") == "gloo":
            # rehearsal only (scripts/bench_ranks_one_gpu.sh): several ranks on ONE GPU, which RCCL refuses
            if not dist.is_initialized():   # (world > 1: made above, before the device check)
                dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        backend = dist.get_backend()

    ksha = kernel_source_sha()
    if args.only_ray_dense:
        print(json.dumps(measure_ray_dense(args.only_ray_dense, args, torch, dev, local_rank, rank, ksha)), flush=True)
        return
    wl = workload(config_id, args.envs, args.layout)
    N = wl["envs"]
    env = make_env(wl, local_rank, rank, args.threads)
    n_obs, n_u = env.n_observations, env.n_u

    # synthetic actions resident in HBM: a ring of distinct batches, uniform in [-1, 1] (counter RNG seeded 1234)
    RING = 64 if N <= 65536 else 16
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    actions = torch.rand((RING, N, n_u), device=dev, generator=gen, dtype=torch.float32) * 2 - 1
    stream = torch.cuda.current_stream().cuda_stream

    # the kernel writes packed rows [obs | reward | done] (float32 [N][n_obs + 2]) straight into this rank's slice
    # of the gather buffer; ONE gather per step, overlapped with the next step's kernel (two buffers)
    from gym_dockauv_amd.parallel import ShardedStepper

    pack_mode = "bf16" if (args.gather_dtype == "bf16" and use_dist) else True

    def step_fn(a, out_local):
        env.step_device(a.data_ptr(), out_local.data_ptr(), stream=stream, packed=pack_mode)

    stepper = ShardedStepper(N, env.packed_row_words(pack_mode), step_fn, dev, world=world, rank=rank, overlap=not args.no_overlap,
                             gather_dtype="bf16" if pack_mode == "bf16" else "f32")
    stepper.use_dist = use_dist
    if pack_mode == "bf16" and args.gather != "rccl":
        raise SystemExit("--gather-dtype bf16 is implemented for the RCCL transport")
    rccl_stepper = stepper
    transport, p2p_note, n_verify = ("rccl" if use_dist else "none"), None, 0
    if use_dist and args.gather in ("auto", "p2p"):
        # peer-to-peer transport: accepted only if every rank mapped its peers AND the gathered rows of the first
        # warm-up steps equal an RCCL all-gather of the same rows bit for bit on every rank
        from gym_dockauv_amd.parallel import P2PShardedStepper
        from gym_dockauv_amd._capi import DockAUVError
        p2p = None
        try:
            p2p = P2PShardedStepper(N, n_obs + 2, step_fn, dev, world=world, rank=rank, overlap=not args.no_overlap)
        except (DockAUVError, ValueError) as e:
            p2p_note = f"p2p set-up failed: {e}"
        if p2p is not None:
            n_verify = max(1, min(8, args.warmup))
            ref = torch.empty((world * N, n_obs + 2), device=dev, dtype=torch.float32)
            ok = 1
            for i in range(n_verify):
                buf = p2p.step(actions[i % RING])
                p2p.wait()
                dist.all_gather_into_tensor(ref, p2p.rows)
                ok &= int(torch.equal(buf.view(torch.int32), ref.view(torch.int32)))
            if not args.no_overlap:
                # the timed region's form: one-call sequences whose gathers ride in the next step kernel; the last two
                # steps' rows are still at hand afterwards, so the ridden gather and the closing one are both compared
                for rep in range(2):
                    i0 = n_verify
                    seq = p2p.make_sequence(env, [actions[i % RING].data_ptr() for i in range(i0, i0 + 5)])
                    p2p.run_sequence(env, seq)
                    n_verify += 5
                    t_last = p2p.gather.t - 1
                    for back in (0, 1):
                        dist.all_gather_into_tensor(ref, p2p.rows2[(t_last - back) & 1])
                        ok &= int(torch.equal(p2p.bufs[(t_last - back) % p2p.gather.nb].view(torch.int32), ref.view(torch.int32)))
            ok &= int(p2p.gather.timed_out() == 0)
            flag = torch.tensor([ok], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                stepper, transport = p2p, "p2p"
            else:
                p2p_note = "p2p rows differed from the RCCL all-gather (or stamps timed out) during warm-up: RCCL used"
                p2p.close()
                p2p = None
        if transport != "p2p" and args.gather == "p2p":
            raise SystemExit(p2p_note)

    # Single GPU, no collective: the K steps of a region are queued by ONE host call (dockauv_step_sequence: K
    # launches of the same kernel, step i reading actions[i % RING]), so that a short kernel is not paced by ~7 us of
    # Python per step.  With RCCL in the loop the steps are issued one by one (the gather sits between them).
    seq_cache = {}

    def run(n, i0=0):
        if not use_dist:
            key = (n, i0 % RING)
            if key not in seq_cache:
                out_ptr = stepper.local_slice(stepper.bufs[0]).data_ptr()
                seq_cache[key] = env.make_step_sequence([actions[i % RING].data_ptr() for i in range(i0, i0 + n)],
                                                        [out_ptr] * n, packed=True)
            env.run_step_sequence(seq_cache[key], stream=stream)
            return
        if transport == "p2p" and not args.no_overlap:
            # one host call: n step kernels, the gather of step t riding in the grid of step kernel t + 1
            key = (n, i0 % RING, stepper.gather.t & 1)
            if key not in seq_cache:
                seq_cache[key] = stepper.make_sequence(env, [actions[i % RING].data_ptr() for i in range(i0, i0 + n)])
            stepper.run_sequence(env, seq_cache[key])
            return
        for i in range(i0, i0 + n):
            stepper.step(actions[i % RING])
        stepper.wait()

    def barrier():
        if use_dist:
            dist.barrier()

    def reduce_max(x):
        if not use_dist:
            return x
        t = torch.tensor([x], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    region_ev_ms = []
    times = timed_regions(run, args.steps, max(0, args.warmup - n_verify), n_verify, args.min_seconds, args.max_reps,
                          torch.cuda.synchronize, barrier, reduce_max, event_ms=None if use_dist else region_ev_ms, torch=torch)
    if transport == "p2p":
        # a stamp that did not arrive within the spin bound voids the regions on every rank: measure again over RCCL
        late = torch.tensor([stepper.gather.timed_out()], device=dev, dtype=torch.int64)
        dist.all_reduce(late, op=dist.ReduceOp.MAX)
        if int(late.item()) != 0:
            if args.gather == "p2p":
                raise SystemExit(f"rank {rank}: peer stamps timed out in the timed region (mask {int(late.item()):#x})")
            p2p_note = f"p2p stamps timed out in the timed region (mask {int(late.item()):#x}): measured again over RCCL"
            stepper.close()
            stepper, transport = rccl_stepper, "rccl"
            seq_cache.clear()
            times = timed_regions(run, args.steps, args.warmup, 0, args.min_seconds, args.max_reps,
                                  torch.cuda.synchronize, barrier, reduce_max)
    dt = statistics.median(times)

    # for comparison: the same steps with the RCCL all-gather as the transport, issued step by step from Python (the
    # p2p regions are queued by one host call: the two paces differ by host pacing as well as by transport)
    rccl_ms = None
    if transport == "p2p" and os.environ.get("DOCKAUV_DIST_BACKEND", "nccl") != "gloo":   # (gloo rehearsal: far too slow)
        k2 = min(args.steps, 2000)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for i in range(k2):
            rccl_stepper.step(actions[i % RING])
        rccl_stepper.wait()
        torch.cuda.synchronize()
        dist.barrier()
        rccl_ms = reduce_max(time.perf_counter() - t0) / k2 * 1e3

    out_l = stepper.rows if transport == "p2p" else stepper.local_slice(stepper.bufs[0])

    # N > 1: the same shard stepped WITHOUT the gather on every rank at the same time (plain step sequences, as the
    # single-GPU line is measured): the per-GPU rate this workload reaches alone.  The weak-scaling efficiency of the
    # line is value / (n_gpus * this) -- the default single-GPU line is a different workload (config 3, the headline).
    alone = None
    if use_dist:
        alone_cache = {}

        def run_alone(n, i0=0):
            key = (n, i0 % RING)
            if key not in alone_cache:
                alone_cache[key] = env.make_step_sequence([actions[i % RING].data_ptr() for i in range(i0, i0 + n)],
                                                          [out_l.data_ptr()] * n, packed=pack_mode)
            env.run_step_sequence(alone_cache[key], stream=stream)

        t_alone = timed_regions(run_alone, args.steps, min(args.warmup, 50), 0, args.min_seconds / 2, args.max_reps,
                                torch.cuda.synchronize, barrier, reduce_max)
        d_alone = statistics.median(t_alone)
        alone = {"per_gpu_value": N * args.steps / d_alone, "unit": "env-steps/s", "ms_per_step": d_alone / args.steps * 1e3,
                 "reps": len(t_alone), "what": "the same shard on every rank at once, no gather (max over ranks)",
                 "weak_scaling_efficiency_of_this_line": (world * N * args.steps / dt) / (world * N * args.steps / d_alone)}

    # the same regions with half-precision rows over the links (RCCL transport): observation columns bfloat16
    bf16_gather = None
    if use_dist and transport == "rccl" and pack_mode is True and os.environ.get("DOCKAUV_DIST_BACKEND", "nccl") != "gloo":
        def step_fn16(a, out_local):
            env.step_device(a.data_ptr(), out_local.data_ptr(), stream=stream, packed="bf16")
        st16 = ShardedStepper(N, env.packed_row_words("bf16"), step_fn16, dev, world=world, rank=rank, overlap=not args.no_overlap,
                              gather_dtype="bf16")
        st16.use_dist = True

        def run16(n, i0=0):
            for i in range(i0, i0 + n):
                st16.step(actions[i % RING])
            st16.wait()
        t16 = timed_regions(run16, args.steps, min(args.warmup, 100), 0, args.min_seconds / 2, args.max_reps,
                            torch.cuda.synchronize, barrier, reduce_max)
        d16 = statistics.median(t16)
        bf16_gather = {"value": world * N * args.steps / d16, "unit": "env-steps/s", "ms_per_step": d16 / args.steps * 1e3,
                       "reps": len(t16), "bytes_per_rank_per_step": st16.bytes_per_rank_per_step,
                       "what": "the same steps with observation columns gathered as bfloat16 (round to nearest even; reward / "
                               "done float32): include/dockauv.h pack_reward_done = 2"}
        del st16

    # roofline of the dominant (only) kernel: per-dispatch start/stop events on the launch stream, cycling through
    # the same action ring as the timed region
    # single GPU: the timed regions themselves, bracketed by HIP events on the launch stream (region time / K); N > 1: the
    # regions contain the gather, so the kernel is timed by per-dispatch events on launches of its own
    n_timed = min(max(args.steps, 256), 1024)
    kernel_us_iso = None
    short_regions = args.steps < 100   # (a region of a few launches is dominated by its first launch after an idle gap)
    if use_dist or args.isolated_kernel_timing or short_regions:
        kernel_us_iso = 0.0
        for i in range(n_timed):
            kernel_us_iso += env.time_steps_device(actions[i % RING].data_ptr(), out_l.data_ptr(), steps=1, stream=stream, packed=pack_mode)
        kernel_us_iso /= n_timed
        torch.cuda.synchronize()
    if use_dist or short_regions:
        kernel_us = kernel_us_iso
        n_timed = (f"hipExtLaunchKernel start/stop events on {n_timed} single dispatches of the step kernel ("
                   + ("the regions of an N > 1 run contain the gather" if use_dist else f"--steps {args.steps}: regions too short to average over") + ")")
    else:
        kernel_us = statistics.median(region_ev_ms) / args.steps * 1e3
        n_timed = len(region_ev_ms) * args.steps
    last = stepper.bufs[0]
    if pack_mode == "bf16":
        o16, r16, d16_ = ShardedStepper.split_bf16(last, n_obs)
        finite = bool(torch.isfinite(o16.float()).all().item() and torch.isfinite(r16).all().item())
        n_done = int(ShardedStepper.split_bf16(out_l, n_obs)[2].sum().item())
    else:
        finite = bool(torch.isfinite(last).all().item())
        n_done = int((out_l[:, n_obs + 1] > 0.5).sum().item())
    closed = closed_loop_rate(env, torch, dev, N, n_obs, n_u, 400) if world == 1 and not use_dist and not args.no_closed_loop else None
    resident = (resident_sequence_rate(env, run, args, torch, config_id, N, kernel_us)
                if world == 1 and not use_dist and not args.no_resident and args.steps >= 2 else None)

    sweep = []
    if world == 1 and not args.no_sweep and not args.envs:
        # the same kernel at batch sizes where the state no longer fits the caches (HBM-relevant roofline points)
        for n_big in args.sweep:
            sweep.append(sweep_point(wl, config_id, n_big, N, args, torch, dev, local_rank, stream, gen))

    if not use_dist:
        collective = "none"
    elif transport == "rccl":
        collective = ("one rccl all_gather of packed [obs|reward|done] per step"
                      + ("" if args.no_overlap else ", overlapped with the next step's kernel"))
    else:
        collective = ("peer-to-peer push of packed [obs|reward|done] into every rank's gather buffer + step stamps "
                      "(dockauv_p2p_*), every rank holds all rows of step t "
                      + ("before step t + 1 starts, issued step by step" if args.no_overlap else
                         "when its step kernel t + 1 has ended: the copy groups of step t ride in the grid of step kernel "
                         "t + 1 (transfer beside the arithmetic, one launch per step, one host call per region)"
                         if stepper._ride_ok else
                         "before step t + 1 starts (gather kernel after each step kernel, one host call per region)")
                      + f"; checked bit-exact against an rccl all_gather during the first {n_verify} warm-up steps")
    devices = [torch.cuda.current_device()]
    if use_dist:
        gathered = [None] * world
        dist.all_gather_object(gathered, {"rank": rank, "local_rank": local_rank, "device": torch.cuda.current_device(),
                                          "name": torch.cuda.get_device_name(local_rank)})
        devices = gathered
    # release the headline env before the sub-results allocate theirs
    if transport == "p2p":
        stepper.close()
    env.close()
    del actions

    if rank == 0:
        out = {
            "metric": "env-steps/sec (batched docking3d)",
            "value": world * N * args.steps / dt,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "reps": len(times),
            "ms_per_step_first_region": times[0] / args.steps * 1e3,
            "ms_per_step_min": min(times) / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": wl["name"], "envs_per_gpu": N, "total_envs": world * N, "n_obs": n_obs, "n_u": n_u,
                       "auto_reset": "in-kernel scenario generation (Philox4x32-10)",
                       "timing": f"median of {len(times)} regions of {args.steps} steps (>= {args.min_seconds} s measured in all), "
                                 "each between barrier + synchronize, max over ranks; open loop: actions from a ring resident in HBM",
                       "collective": collective,
                       "gather_dtype": ("bf16" if pack_mode == "bf16" else "f32") if use_dist else None,
                       "gather_bytes_per_rank_per_step": (stepper.bytes_per_rank_per_step if hasattr(stepper, "bytes_per_rank_per_step")
                                                          else N * (n_obs + 2) * 4) if use_dist else 0,
                       "world_size": (dist.get_world_size() if use_dist else 1), "backend": backend, "ranks": devices,
                       **({"gather_note": p2p_note} if p2p_note else {}),
                       **({"rccl_all_gather_ms_per_step_python_issued": rccl_ms} if rccl_ms is not None else {}),
                       "obs_finite": finite, "done_last_step_rank0": n_done,
                       "kernel_source_sha": ksha},
            "roofline": roofline_of(config_id, N, kernel_us, n_timed, ksha),
        }
        if kernel_us_iso is not None and not use_dist:
            out["kernel_us_isolated_launches"] = kernel_us_iso
        if closed:
            out["closed_loop"] = closed
        if resident:
            out["sequence_resident"] = resident
        if alone:
            out["same_workload_without_gather"] = alone
        if bf16_gather:
            out["bf16_gather"] = bf16_gather
        if use_dist:
            # what the gather asks of the fabric: every rank sends its rows to each of the N - 1 others once per step
            # (xGMI is point to point: one link per peer, ~64 GB/s per direction at its 153 GB/s bidirectional peak)
            bpr = out["config"]["gather_bytes_per_rank_per_step"]
            step_s = dt / args.steps
            out["link_GBps_achieved"] = {
                "per_rank_egress": bpr * (world - 1) / step_s / 1e9, "per_link_direction": bpr / step_s / 1e9,
                "what": "gathered bytes per rank and step x (N - 1) peers / step time of this line (per link and direction: "
                        "one rank's rows / step time); if the step time equals same_workload_without_gather's, the links "
                        "are not what bounds it",
                **({"bf16_per_rank_egress": bf16_gather["bytes_per_rank_per_step"] * (world - 1) / (bf16_gather["ms_per_step"] * 1e-3) / 1e9}
                   if bf16_gather else {})}
        if sweep:
            out["sweep"] = sweep
        if world == 1 and not use_dist and not args.no_configs and not args.envs:
            subs = []
            for cid in (2, 3, 4, 5):
                if cid == config_id:
                    subs.append({"workload": wl["name"], "envs": N, "value": out["value"], "unit": "env-steps/s",
                                 "ms_per_step": out["ms_per_step"], "reps": len(times), "kernel_us": kernel_us,
                                 "roofline": out["roofline"], "closed_loop": closed, "sequence_resident": resident, "headline": True})
                else:
                    subs.append(measure_single(cid, 0, args, torch, dev, local_rank, rank, ksha))
                if cid == 5:     # both layouts of the mixed batch
                    subs[-1]["layout"] = "interleaved"
                    subs.append(measure_single(5, 0, args, torch, dev, local_rank, rank, ksha, layout="vehicle_sorted"))
                    subs[-1]["layout"] = "vehicle_sorted"
            for cid in (3, 4):   # the ray stage where a docking policy lives
                subs.append(measure_ray_dense(cid, args, torch, dev, local_rank, rank, ksha))
            if not args.no_cpu:  # the CPU path on the SAME scenario / vehicle / fan / step size, per config (a few seconds each)
                for sub in subs:
                    cid = int(sub["workload"][6])
                    if cid != config_id and "ray-dense" not in sub["workload"] and sub.get("layout") != "vehicle_sorted":
                        sub["cpu_baseline"] = cpu_baseline(workload(cid, 2), max(2.0, args.cpu_seconds / 4))
            out["configs"] = subs
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(wl, args.cpu_seconds)
            allv, procs = cpu_baseline_all_cores(config_id, args.cpu_seconds)
            out["cpu_baseline"]["all_cores_value"] = allv        # one oracle process per CPU of this host
            out["cpu_baseline"]["all_cores"] = procs
        sys.stdout.flush()
        if saved_stdout is not None:
            os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        if saved_stdout is not None:
            os.dup2(2, 1)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
