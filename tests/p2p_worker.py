"""Worker of tests/test_gpu_p2p.py: one rank of a peer-to-peer gather rehearsal.  All ranks use the box's ONE GPU
(the IPC mapping, the push kernel, the stamps and the bounded wait are the same code that runs across GPUs; what the
rehearsal cannot show is the fabric itself).  Usage: python -m tests.p2p_worker RANK WORLD PORT N_LOCAL STEPS"""
import os
import sys

import numpy as np


def main() -> int:
    rank, world, port, n_local, steps = (int(x) for x in sys.argv[1:6])
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
    if os.environ.get("P2P_WORKER_MODE") == "stall":
        # a peer that stops stepping: the surviving rank's ShardedTorchDocking3d.step must RAISE (not hand out stale rows)
        from gym_dockauv_amd._capi import DockAUVError
        from gym_dockauv_amd.envs.torch_env import ShardedTorchDocking3d
        try:
            sh = ShardedTorchDocking3d(BASE_CONFIG, num_envs=world * n_local, scenario="SimpleCurrentDocking3d", device=0,
                                       transport="p2p", device_seed=900, host_seed=70, check_every=8, p2p_max_spins=200_000)
            assert sh.transport == "p2p", sh.transport_note
            sh.reset()
            a = torch.zeros((world * n_local, sh.n_u), device=dev)
            raised = False
            n_steps = steps if rank == 0 else 5          # rank 1 (and above) stop after five steps
            try:
                for t in range(n_steps):
                    sh.step(a)
            except DockAUVError as e:
                raised = True
                print(f"rank {rank}: step raised as it must: {e}", flush=True)
            if rank == 0:
                assert raised, "rank 0 kept stepping on stale rows"
            else:
                import time
                time.sleep(3.0)                          # long enough for rank 0 to run out of spins
            # (no collective from here on: the job is broken by design; each rank releases its own resources)
            sh.batch.close()
            print(f"rank {rank}: stall rehearsal done, 0 bad", flush=True)
            os._exit(0)
        finally:
            pass
    try:
        # every rank also steps private copies of the OTHER ranks' shards (same seeds -> same rows, the kernel is
        # deterministic): the gathered buffer must equal their concatenation bit for bit
        envs = []
        for r in range(world):
            e = BatchedDocking3d(BASE_CONFIG, num_envs=n_local, scenario="SimpleCurrentDocking3d", device=0, precision="f32",
                                 reset_mode="device", device_seed=77 + r, rng="batched")
            e._gen = np.random.default_rng(5 + r)
            e.reset()
            envs.append(e)
        row = envs[0].n_observations + 2
        n_u = envs[0].n_u
        gen = torch.Generator(device=dev)
        gen.manual_seed(99)
        actions = torch.rand((steps, world, n_local, n_u), device=dev, generator=gen) * 2 - 1   # same on every rank
        mirror = torch.zeros((world, n_local, row), device=dev)
        bad = 0
        combos = ((False, True), (True, True), (False, False), (True, False))
        order = [int(k) for k in os.environ.get("P2P_WORKER_ORDER", "0,1,2,3").split(",")]
        for overlap, fused in [combos[k] for k in order]:     # overlap = lag 1 here
            def step_fn(a, out):
                envs[rank].step_device(a.data_ptr(), out.data_ptr(), stream=stream, packed=True)
            st = P2PShardedStepper(n_local, row, step_fn, dev, world=world, rank=rank, lag=1 if overlap else 0, fused=fused)
            expected = []
            got = []
            for t in range(steps):
                buf = st.step(actions[t, rank])
                for r in range(world):
                    if r != rank:
                        envs[r].step_device(actions[t, r].data_ptr(), mirror[r].data_ptr(), stream=stream, packed=True)
                mirror[rank].copy_(st.rows)
                expected.append(mirror.reshape(world * n_local, row).clone())
                if not overlap:
                    got.append(buf.clone())              # lag 0: complete as soon as the step's wait has run
                elif t > 0:
                    got.append(prev.clone())             # lag 1: step t's wait covers step t - 1
                prev = buf
            st.wait()
            if overlap:
                got.append(prev.clone())
            assert st.gather.timed_out() == 0, f"rank {rank}: stamps timed out, mask {st.gather.timed_out():#x}"
            for t in range(steps):
                if not torch.equal(got[t].view(torch.int32), expected[t].view(torch.int32)):
                    bad += 1
                    print(f"rank {rank} overlap {overlap} step {t}: gathered rows differ", flush=True)
            # open-loop sequences (one host call each), on one stream and on two: afterwards every buffer of the
            # rotation holds the gathered rows of its step
            for two, ride in ((False, True), (False, False), (True, False)):
                seq = st.make_sequence(envs[rank], [actions[t, rank].data_ptr() for t in range(steps)])
                t_first = st.gather.t
                st.run_sequence(envs[rank], seq, two_streams=two, ride=ride)
                st.wait()
                exp_seq = []
                for t in range(steps):
                    for r in range(world):
                        if r != rank:
                            envs[r].step_device(actions[t, r].data_ptr(), mirror[r].data_ptr(), stream=stream, packed=True)
                    exp_seq.append(mirror.clone())
                torch.cuda.synchronize()
                for t in range(steps - st.gather.nb, steps):
                    buf = st.bufs[(t_first + t) % st.gather.nb].reshape(world, n_local, row)
                    for r in range(world):
                        if r != rank and not torch.equal(buf[r].view(torch.int32), exp_seq[t][r].view(torch.int32)):
                            bad += 1
                            print(f"rank {rank} sequence (two streams: {two}, ride: {ride}) step {t}: rows of rank {r} differ", flush=True)
                if not torch.equal(st.local_slice(st.bufs[(t_first + steps - 1) % st.gather.nb]), st.rows):
                    bad += 1
                    print(f"rank {rank} sequence (two streams: {two}, ride: {ride}): own rows differ", flush=True)
                # the buffers were read from the HOST, four steps late: no rank may go on (and reuse them) before
                # every rank has looked (a learner reads step t on the step stream before its step t + 2)
                dist.barrier()
            # pace of the whole per-step chain (step kernel, push, stamps) with every rank on this one GPU
            torch.cuda.synchronize()
            dist.barrier()
            import time
            t0 = time.perf_counter()
            for t in range(200):
                st.step(actions[t % steps, rank])
            st.wait()
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / 200 * 1e6
            assert st.gather.timed_out() == 0
            for t in range(200):              # keep the private copies of the other shards in step
                for r in range(world):
                    if r != rank:
                        envs[r].step_device(actions[t % steps, r].data_ptr(), mirror[r].data_ptr(), stream=stream, packed=True)
            us2 = {}
            for two in (False, True, "ride"):
                seq = st.make_sequence(envs[rank], [actions[t % steps, rank].data_ptr() for t in range(200)])
                torch.cuda.synchronize()
                dist.barrier()
                t0 = time.perf_counter()
                st.run_sequence(envs[rank], seq, two_streams=two is True, ride=two == "ride")
                st.wait()
                torch.cuda.synchronize()
                us2[two] = (time.perf_counter() - t0) / 200 * 1e6
                assert st.gather.timed_out() == 0
                for t in range(200):
                    for r in range(world):
                        if r != rank:
                            envs[r].step_device(actions[t % steps, r].data_ptr(), mirror[r].data_ptr(), stream=stream, packed=True)
            print(f"rank {rank} overlap {overlap} fused {fused}: {us:.1f} us per step+gather from Python; one-call sequence "
                  f"{us2['ride']:.1f} us (gathers riding in the next step kernel), {us2[False]:.1f} us (gather kernels, one stream), {us2[True]:.1f} us (two streams) ({world} ranks on one GPU, {n_local} envs each)", flush=True)
            dist.barrier()
            st.close()
        for e in envs:
            e.close()
        # closed-loop learner surface: ShardedTorchDocking3d.step returns ALL ranks' rows; a toy policy closes the loop
        from gym_dockauv_amd.envs.torch_env import ShardedTorchDocking3d
        sh = ShardedTorchDocking3d(BASE_CONFIG, num_envs=world * n_local, scenario="SimpleCurrentDocking3d", device=0,
                                   transport="p2p", device_seed=900, host_seed=70)
        mir = []
        for r in range(world):
            e = BatchedDocking3d(BASE_CONFIG, num_envs=n_local, scenario="SimpleCurrentDocking3d", device=0, precision="f32",
                                 reset_mode="device", device_seed=900 + r, rng="batched")
            e._gen = np.random.default_rng(70 + r)
            e.reset()
            mir.append(e)
        obs = sh.reset()
        W = torch.linspace(-1, 1, sh.n_obs * sh.n_u, device=dev).reshape(sh.n_obs, sh.n_u)
        exp = torch.zeros((world, n_local, row), device=dev)
        exp_obs = torch.zeros((world * n_local, sh.n_obs), device=dev)
        for t in range(steps):
            a = torch.tanh(obs @ W + 0.1 * t).contiguous()              # same on every rank: obs is global
            a_exp = torch.tanh(exp_obs @ W + 0.1 * t).contiguous()
            obs, rew, done = sh.step(a)
            for r in range(world):
                mir[r].step_device(a_exp[r * n_local:(r + 1) * n_local].contiguous().data_ptr(), exp[r].data_ptr(), stream=stream, packed=True)
            e2 = exp.reshape(world * n_local, row)
            exp_obs = e2[:, :sh.n_obs].clone()
            if not (torch.equal(obs.view(torch.int32), e2[:, :sh.n_obs].contiguous().view(torch.int32))
                    and torch.equal(rew, e2[:, sh.n_obs]) and torch.equal(done, e2[:, sh.n_obs + 1] > 0.5)):
                bad += 1
                print(f"rank {rank} sharded env step {t}: global rows differ", flush=True)
        assert sh.stepper.gather.timed_out() == 0
        dist.barrier()
        sh.close()
        for e in mir:
            e.close()
        print(f"rank {rank}: gathered steps checked, {bad} bad", flush=True)
        return 1 if bad else 0
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
