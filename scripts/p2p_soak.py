"""Soak of the riding gather: WORLD rank processes on the box's one GPU run many short open-loop regions
(dockauv_step_gather_sequence, gathers riding in the next step kernel) and compare, after every region, the gathered
rows of its last two steps with privately stepped copies of the other ranks' shards, bit for bit.
usage (one process per rank): python scripts/p2p_soak.py RANK WORLD PORT [REGIONS] [STEPS_PER_REGION] [N_LOCAL]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main() -> int:
    rank, world, port = (int(x) for x in sys.argv[1:4])
    regions = int(sys.argv[4]) if len(sys.argv) > 4 else 2000
    steps = int(sys.argv[5]) if len(sys.argv) > 5 else 33          # odd: the row-buffer parity alternates
    n_local = int(sys.argv[6]) if len(sys.argv) > 6 else 1024
    import torch
    import torch.distributed as dist
    from gym_dockauv_amd.envs.batched import BASE_CONFIG, BatchedDocking3d
    from gym_dockauv_amd.parallel import P2PShardedStepper
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    envs = []
    for r in range(world):
        e = BatchedDocking3d(BASE_CONFIG, num_envs=n_local, scenario="SimpleCurrentDocking3d", device=0, precision="f32",
                             reset_mode="device", device_seed=1000 + r, rng="batched")
        e._gen = np.random.default_rng(50 + r)
        e.reset()
        envs.append(e)
    row, n_u = envs[0].n_observations + 2, envs[0].n_u
    gen = torch.Generator(device=dev)
    gen.manual_seed(4242)
    actions = torch.rand((64, world, n_local, n_u), device=dev, generator=gen) * 2 - 1
    st = P2PShardedStepper(n_local, row, lambda a, o: None, dev, world=world, rank=rank)
    mirror = torch.zeros((2, world, n_local, row), device=dev)
    seqs = {}
    bad = 0
    t_start = time.perf_counter()
    for reg in range(regions):
        i0 = (reg * steps) % 64
        key = (i0, st.gather.t & 1)
        if key not in seqs:
            seqs[key] = st.make_sequence(envs[rank], [actions[(i0 + t) % 64, rank].data_ptr() for t in range(steps)])
        st.run_sequence(envs[rank], seqs[key], ride=True)
        for t in range(steps):
            for r in range(world):
                if r != rank:
                    envs[r].step_device(actions[(i0 + t) % 64, r].data_ptr(), mirror[t & 1, r].data_ptr(), stream=stream, packed=True)
        torch.cuda.synchronize()
        t_last = st.gather.t - 1
        for back in (0, 1):
            buf = st.bufs[(t_last - back) % st.gather.nb].reshape(world, n_local, row)
            exp = mirror[(steps - 1 - back) & 1]
            for r in range(world):
                ok = torch.equal(buf[r].view(torch.int32), (st.rows2[(t_last - back) & 1] if r == rank else exp[r]).view(torch.int32))
                if not ok:
                    bad += 1
                    if bad < 10:
                        print(f"rank {rank} region {reg} step -{back}: rows of rank {r} differ", flush=True)
        dist.barrier()     # host-side readers: nobody reuses the buffers before every rank has looked
        if reg % 500 == 499:
            print(f"rank {rank}: {reg + 1} regions, {bad} bad, {time.perf_counter() - t_start:.1f} s", flush=True)
    late = st.gather.timed_out()
    print(f"rank {rank}: {regions * steps} steps in {regions} regions, {bad} mismatching slices, time-out mask {late:#x}", flush=True)
    st.close()
    for e in envs:
        e.close()
    dist.destroy_process_group()
    return 1 if (bad or late) else 0


if __name__ == "__main__":
    sys.exit(main())
