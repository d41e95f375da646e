"""Closed-loop rollout over several GPUs with a single (replicated) toy policy: every rank steps its shard and gets ALL
ranks' observations back (ShardedTorchDocking3d), computes the same actions from them and uses its own rows.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/sharded_rollout.py \\
        [--envs-per-gpu 4096] [--steps 2000] [--scenario ObstaclesCurrentDocking3d] [--transport p2p|rccl]
(one rank: plain `python scripts/sharded_rollout.py`)"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--scenario", default="SimpleCurrentDocking3d")
    ap.add_argument("--transport", choices=["p2p", "rccl"], default="p2p")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from gym_dockauv_amd.config.env_config import TRAIN_CONFIG
    from gym_dockauv_amd.envs.torch_env import ShardedTorchDocking3d
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    env = ShardedTorchDocking3d(TRAIN_CONFIG, num_envs=world * args.envs_per_gpu, scenario=args.scenario, device=local_rank,
                                transport=args.transport, device_seed=1, host_seed=1)
    torch.manual_seed(0)                                   # the same policy on every rank
    W = (torch.randn((env.n_obs, env.n_u), device=dev) * 0.3).contiguous()
    obs = env.reset()
    ret = torch.zeros(env.num_envs, device=dev)
    n_done = torch.zeros((), device=dev)
    for t in range(args.steps + 100):
        if t == 100:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        actions = 0.2 * torch.tanh(obs @ W)                    # [num_envs, n_u]; env.step uses this rank's rows
        obs, reward, done = env.step(actions)
        ret += reward
        n_done += done.sum()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        print(f"{world} rank(s) x {args.envs_per_gpu} envs, {args.scenario}, transport {args.transport}: "
              f"{env.num_envs * args.steps / dt:.3e} env-steps/s closed loop ({dt / args.steps * 1e6:.1f} us per step incl. policy), "
              f"{int(n_done.item())} episodes finished, mean return per env {ret.mean().item():.2f}", flush=True)
    env.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
