#!/usr/bin/env python3
"""
bench.py -- env-steps/s of the batched docking3d step() on N MI355X (BASELINE.json metric), one process per GPU.

  python bench.py --gpus 1 --steps K --warmup W                       (defaults finish in about a minute)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

A "step" is one launch of the fused HIP step kernel over this rank's envs (weak scaling: the per-GPU env count is
fixed; at N = 1 the K launches of a region are queued by one dockauv_step_sequence call), followed -- for N > 1 --
by one RCCL all-gather of [obs | reward | done] over xGMI so that a single learner
sees all observations.  Inputs (actions) are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

Workloads (--config):
  2  BlueROV2, SimpleDocking3d, 4 096 envs/GPU, no obstacles (pure 6-DOF RKF45 + reward)      [default]
  3  BlueROV2, 16-beam fan vs 8 spheres, 65 536 envs/GPU
  4  LAUV, ObstaclesDocking3d (5 capsules, 63 rays), t_step_size 0.02, 32 768 envs/GPU
  5  BlueROV2/LAUV 50/50, ObstaclesCurrentDocking3d, t_step_size 0.02, 65 536 envs/GPU
"""
import argparse
import copy
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES = {2: 356, 3: 420, 4: 460, 5: 482}   # algorithmic bytes per env-step
HBM_PEAK_GBPS = 8000.0                           # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def workload(config_id: int, envs: int):
    from gym_dockauv_amd.config.env_config import BASE_CONFIG
    cfg = copy.deepcopy(BASE_CONFIG)
    fan16 = {"alpha": 30 * np.pi / 180, "beta": 30 * np.pi / 180, "ray_per_deg": 10 * np.pi / 180}
    if config_id == 2:
        return dict(cfg=cfg, scenario="SimpleDocking3d", envs=envs or 4096, vehicles=None,
                    name="config2: BlueROV2 SimpleDocking3d, no sensors, h=0.1")
    if config_id == 3:
        cfg["radar"].update(fan16)
        return dict(cfg=cfg, scenario="SphereDocking3d", envs=envs or 65536, vehicles=None,
                    name="config3: BlueROV2 + 16-beam fan vs 8 spheres, h=0.1")
    if config_id == 4:
        cfg["vehicle"] = "LAUV"
        cfg["t_step_size"] = 0.02
        return dict(cfg=cfg, scenario="ObstaclesDocking3d", envs=envs or 32768, vehicles=None,
                    name="config4: LAUV ObstaclesDocking3d (5 capsules, 63 rays), h=0.02")
    if config_id == 5:
        cfg["t_step_size"] = 0.02
        n = envs or 65536
        if os.environ.get("DOCKAUV_CONFIG5_SORTED") == "1":
            # vehicle-sorted layout: first half BlueROV2, second half LAUV -> every wave but one is homogeneous
            return dict(cfg=cfg, scenario="ObstaclesCurrentDocking3d", envs=n,
                        vehicles=["BlueROV2"] * (n // 2) + ["LAUV"] * (n - n // 2),
                        name="config5: BlueROV2/LAUV vehicle-sorted, ObstaclesCurrentDocking3d, h=0.02")
        return dict(cfg=cfg, scenario="ObstaclesCurrentDocking3d", envs=n,
                    vehicles=["BlueROV2" if i % 2 == 0 else "LAUV" for i in range(n)],
                    name="config5: BlueROV2/LAUV interleaved, ObstaclesCurrentDocking3d, h=0.02")
    raise SystemExit(f"unknown --config {config_id}")


def cpu_baseline(wl, seconds: float):
    """The NumPy oracle (a port of the reference's per-env NumPy path) timed on this host, one core, on a bounded
    sample of the same workload.  Reported next to the GPU number; it is NOT on the product path."""
    from oracle import dockauv_oracle as orc
    scenario = wl["scenario"] if wl["scenario"] in orc.SCENARIOS else "SimpleDocking3d"
    veh = wl["cfg"]["vehicle"]
    env = orc.OracleEnv(scenario, {"vehicle": veh, "t_step_size": wl["cfg"]["t_step_size"],
                                   "radar": {k: wl["cfg"]["radar"][k] for k in ("alpha", "beta", "ray_per_deg", "max_dist")}})
    env.reset(seed=0)
    n_u = env.model.n_u
    rs = np.random.RandomState(1234)
    for _ in range(50):
        _, _, d, _ = env.step(rs.uniform(-1, 1, n_u))
        if d:
            env.reset()
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(100):
            _, _, d, _ = env.step(rs.uniform(-1, 1, n_u))
            if d:
                env.reset()
        n += 100
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"oracle/dockauv_oracle.py (NumPy, float64, 1 env, 1 core), {n} steps of {scenario}/{veh} "
                      f"in {dt:.1f} s, uniform random actions, reset on done"}


def cpu_baseline_all_cores(args, seconds: float):
    """The same oracle loop in one child process per CPU of this host (independent envs, as SURVEY.md section 8d asks);
    children never touch the GPU.  Returns (aggregate env-steps/s, processes)."""
    import subprocess
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    procs = max(1, min(avail, 16))   # a one-GPU box's CPU share is 16 cores
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", "--config", str(args.config),
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)     # 0.12 s at 6 us per step: short hiccups of a box average out
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--envs", type=int, default=0, help="envs per GPU (0 = the config's size)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-worker", action="store_true", help=argparse.SUPPRESS)   # child of cpu_baseline_all_cores
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--no-overlap", action="store_true", help="do not overlap the all-gather with the next kernel")
    ap.add_argument("--gather", choices=["auto", "rccl", "p2p"], default="auto",
                    help="N > 1: transport of the per-step gather.  auto = peer-to-peer push (dockauv_p2p_*) if it maps "
                         "and reproduces the RCCL all-gather bit for bit during warm-up, else RCCL")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--sweep", type=int, nargs="*", default=[65536, 262144, 1048576])
    args = ap.parse_args()
    if args.cpu_worker:      # oracle loop only: no torch, no GPU
        print(cpu_baseline(workload(args.config, 1), args.cpu_seconds)["value"])
        return

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("DOCKAUV_FORCE_DIST") == "1"   # the latter: 1-rank rehearsal of the RCCL path
    saved_stdout = None
    if use_dist:
        # RCCL prints a version banner on stdout when the communicator is created: keep stdout for the ONE JSON line
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if os.environ.get("DOCKAUV_DIST_BACKEND", "nccl") == "gloo":
            # rehearsal only (scripts/bench_ranks_one_gpu.sh): several ranks on ONE GPU, which RCCL refuses
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

    from gym_dockauv_amd.envs.batched import BatchedDocking3d
    wl = workload(args.config, args.envs)
    N = wl["envs"]
    env = BatchedDocking3d(wl["cfg"], num_envs=N, scenario=wl["scenario"], device=local_rank, precision="f32",
                           reset_mode="device", device_seed=0x5EED0000 + rank, rng="batched", vehicles=wl["vehicles"],
                           threads_per_group=args.threads)
    env._gen = np.random.default_rng(1000 + rank)
    env.reset()
    n_obs, n_u = env.n_observations, env.n_u

    # synthetic actions resident in HBM: a ring of distinct batches, uniform in [-1, 1] (counter RNG seeded 1234)
    RING = 64 if N <= 65536 else 16
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    actions = torch.rand((RING, N, n_u), device=dev, generator=gen, dtype=torch.float32) * 2 - 1
    stream = torch.cuda.current_stream().cuda_stream

    # the kernel writes packed rows [obs | reward | done] (float32 [N][n_obs + 2]) straight into this rank's slice
    # of the gather buffer; ONE all-gather per step, overlapped with the next step's kernel (two buffers)
    from gym_dockauv_amd.parallel import ShardedStepper

    def step_fn(a, out_local):
        env.step_device(a.data_ptr(), out_local.data_ptr(), stream=stream, packed=True)

    stepper = ShardedStepper(N, n_obs + 2, step_fn, dev, world=world, rank=rank, overlap=not args.no_overlap)
    stepper.use_dist = use_dist
    rccl_stepper = stepper
    transport, p2p_note, n_verify = ("rccl" if use_dist else "none"), None, 0
    two_streams = False     # gathers on a second stream: measured slower on one GPU at every size (parallel.py)
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
    # launches of the same kernel, step i reading actions[i % RING]), so that a 6 us kernel is not paced by ~7 us of
    # Python per step.  With a collective in the loop the steps are issued one by one (the gather sits between them).
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
            # one host call: n step kernels on the compute stream, their gathers on a second stream
            key = (n, i0 % RING, stepper.gather.t & 1)
            if key not in seq_cache:
                seq_cache[key] = stepper.make_sequence(env, [actions[i % RING].data_ptr() for i in range(i0, i0 + n)])
            stepper.run_sequence(env, seq_cache[key])
            return
        for i in range(i0, i0 + n):
            stepper.step(actions[i % RING])
        stepper.wait()

    def timed_region(n_warm, i_warm):
        run(n_warm, i_warm)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(args.steps, args.warmup)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt_], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_ = float(t.item())
        return dt_

    dt = timed_region(max(0, args.warmup - n_verify), n_verify)
    if transport == "p2p":
        # a stamp that did not arrive within the spin bound voids the region on every rank: measure again over RCCL
        late = torch.tensor([stepper.gather.timed_out()], device=dev, dtype=torch.int64)
        dist.all_reduce(late, op=dist.ReduceOp.MAX)
        if int(late.item()) != 0:
            if args.gather == "p2p":
                raise SystemExit(f"rank {rank}: peer stamps timed out in the timed region (mask {int(late.item()):#x})")
            p2p_note = f"p2p stamps timed out in the timed region (mask {int(late.item()):#x}): measured again over RCCL"
            stepper.close()
            stepper, transport = rccl_stepper, "rccl"
            seq_cache.clear()
            dt = timed_region(args.warmup, 0)

    # for comparison: the same steps with the RCCL all-gather as the transport (short region, reported beside `value`)
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
        t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        rccl_ms = float(t.item()) / k2 * 1e3

    # roofline of the dominant (only) kernel: per-dispatch start/stop events on the launch stream, cycling through
    # the same action ring as the timed region
    out_l = stepper.rows if transport == "p2p" else stepper.local_slice(stepper.bufs[0])
    k_steps = min(args.steps, 1024)
    kernel_us = 0.0
    for i in range(k_steps):
        kernel_us += env.time_steps_device(actions[i % RING].data_ptr(), out_l.data_ptr(), steps=1, stream=stream, packed=True)
    n_timed = k_steps
    kernel_us /= n_timed
    torch.cuda.synchronize()
    last = stepper.bufs[0]
    finite = bool(torch.isfinite(last).all().item())
    n_done = int((out_l[:, n_obs + 1] > 0.5).sum().item())

    sweep = []
    if world == 1 and not args.no_sweep and not args.envs:
        # the same kernel at batch sizes where the state no longer fits the caches (HBM-relevant roofline points)
        for n_big in args.sweep:
            e2 = BatchedDocking3d(wl["cfg"], num_envs=n_big, scenario=wl["scenario"], device=local_rank, precision="f32",
                                  reset_mode="device", device_seed=0xABC, rng="batched",
                                  vehicles=(wl["vehicles"] * (n_big // N + 1))[:n_big] if wl["vehicles"] else None)
            e2._gen = np.random.default_rng(7)
            e2.reset()
            a2 = torch.rand((4, n_big, n_u), device=dev, generator=gen, dtype=torch.float32) * 2 - 1
            o2 = torch.zeros((n_big, n_obs + 2), device=dev, dtype=torch.float32)
            for r in range(8):
                e2.step_device(a2[r % 4].data_ptr(), o2.data_ptr(), stream=stream, packed=True)
            torch.cuda.synchronize()
            us = sum(e2.time_steps_device(a2[r % 4].data_ptr(), o2.data_ptr(), steps=10, stream=stream, packed=True) for r in range(8)) / 8
            torch.cuda.synchronize()
            gbps = ALGO_BYTES[args.config] * n_big / (us * 1e-6) / 1e9
            sweep.append({"envs": n_big, "kernel_us": us, "env_steps_per_s_kernel": n_big / (us * 1e-6),
                          "achieved_GBps": gbps, "frac_of_8TBps": gbps / HBM_PEAK_GBPS})
            e2.close()
            del a2, o2

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
    if rank == 0:
        bytes_per_launch = ALGO_BYTES[args.config] * N
        achieved = bytes_per_launch / (kernel_us * 1e-6) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(f"config{args.config}_envs{N}")
            except Exception:
                traffic = None
        out = {
            "metric": "env-steps/sec (batched docking3d)",
            "value": world * N * args.steps / dt,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": wl["name"], "envs_per_gpu": N, "total_envs": world * N, "n_obs": n_obs, "n_u": n_u,
                       "auto_reset": "in-kernel scenario generation (Philox4x32-10)",
                       "collective": collective,
                       **({"gather_note": p2p_note} if p2p_note else {}),
                       **({"rccl_all_gather_ms_per_step": rccl_ms} if rccl_ms is not None else {}),
                       "obs_finite": finite, "done_last_step_rank0": n_done},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": "dockauv::step_kernel", "kernel_us": kernel_us,
                         "algorithmic_bytes_per_env_step": ALGO_BYTES[args.config], "envs_per_launch": N,
                         "timing": f"hipExtLaunchKernel start/stop events on the launch stream, {n_timed} launches after the timed region"},
        }
        if sweep:
            out["sweep"] = sweep
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(wl, args.cpu_seconds)
            allv, procs = cpu_baseline_all_cores(args, args.cpu_seconds)
            out["cpu_baseline"]["all_cores_value"] = allv        # one oracle process per CPU of this host
            out["cpu_baseline"]["all_cores"] = procs
        sys.stdout.flush()
        if saved_stdout is not None:
            os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        if saved_stdout is not None:
            os.dup2(2, 1)
    if transport == "p2p":
        stepper.close()
    env.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
