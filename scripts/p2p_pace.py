"""One rank, one GPU: cost of the gather chain next to the bare step kernel at BASELINE's config-2 batch (the local
part of the multi-GPU step: launches, the local copy, stamps; the fabric adds the transfer).  gpurun box."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from gym_dockauv_amd.envs.batched import BASE_CONFIG, BatchedDocking3d   # noqa: E402
from gym_dockauv_amd.parallel import P2PShardedStepper   # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
env = BatchedDocking3d(BASE_CONFIG, num_envs=N, scenario="SimpleDocking3d", device=0, precision="f32", reset_mode="device",
                       device_seed=1, rng="batched")
env._gen = np.random.default_rng(1)
env.reset()
stream = torch.cuda.current_stream().cuda_stream
actions = torch.rand((64, N, env.n_u), device=dev) * 2 - 1
row = env.n_observations + 2


def step_fn(a, out):
    env.step_device(a.data_ptr(), out.data_ptr(), stream=stream, packed=True)


K = 2000
for overlap in (False, True):
    st = P2PShardedStepper(N, row, step_fn, dev, world=1, rank=0, overlap=overlap)
    plain = env.make_step_sequence([actions[i % 64].data_ptr() for i in range(K)], [st.rows2[0].data_ptr()] * K, packed=True)
    env.run_step_sequence(plain, stream=stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    env.run_step_sequence(plain, stream=stream)
    torch.cuda.synchronize()
    base = (time.perf_counter() - t0) / K * 1e6
    res = {}
    for two in (False, True, "ride"):
        seq = st.make_sequence(env, [actions[i % 64].data_ptr() for i in range(K)])
        st.run_sequence(env, seq, two_streams=two is True, ride=two == "ride")
        st.wait()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st.run_sequence(env, seq, two_streams=two is True, ride=two == "ride")
        st.wait()
        torch.cuda.synchronize()
        res[two] = (time.perf_counter() - t0) / K * 1e6
    print(f"{N} envs, lag {int(overlap)}: step kernel alone {base:.2f} us/step; + gather kernel, one stream {res[False]:.2f}; "
          f"two streams {res[True]:.2f}; gathers riding in the next step kernel {res['ride']:.2f}", flush=True)
    st.close()
env.close()
