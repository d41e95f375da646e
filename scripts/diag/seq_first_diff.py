"""Where a resident step sequence first differs from single launches (diagnostic of tests/test_gpu_reset.py::
test_step_sequence_equals_single_steps): python scripts/diag/seq_first_diff.py <config> <envs> <threads> [max_timesteps]"""
import copy
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from gym_dockauv_amd.envs.batched import BatchedDocking3d

cid, N, threads = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mt = int(sys.argv[4]) if len(sys.argv) > 4 else 23
K = 70
dev = torch.device("cuda", 0)
wl = bench.workload(cid, N)
cfg = copy.deepcopy(wl["cfg"])
cfg["max_timesteps"] = mt
outs = {}
for mode in ("single", "resident"):
    env = BatchedDocking3d(cfg, num_envs=N, scenario=wl["scenario"], precision="f32", reset_mode="device", device_seed=99,
                           rng="batched", vehicles=wl["vehicles"], threads_per_group=threads)
    env._gen = np.random.default_rng(3)
    env.reset()
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    acts = torch.rand((K, N, env.n_u), device=dev, generator=g) * 2 - 1
    out = torch.zeros((K, N, env.packed_row_words(True)), device=dev, dtype=torch.float32)
    stream = torch.cuda.current_stream().cuda_stream
    if mode == "single":
        for k in range(K):
            env.step_device(acts[k].data_ptr(), out[k].data_ptr(), stream=stream, packed=True)
    else:
        env.set_sequence_resident(True)
        ios = env.make_step_sequence([acts[k].data_ptr() for k in range(K)], [out[k].data_ptr() for k in range(K)], packed=True)
        env.run_step_sequence(ios, stream=stream)
    torch.cuda.synchronize()
    env.synchronize()
    outs[mode] = out.cpu().numpy()
    vid = np.asarray(env.vehicle_id).copy() if hasattr(env, "vehicle_id") else None
    env.close()
a, b = outs["single"], outs["resident"]
n_obs = a.shape[2] - 2
diff = (a.view(np.uint32) != b.view(np.uint32))
print("rows differing per step:", diff.any(axis=2).sum(axis=1).tolist())
ks = np.nonzero(diff.any(axis=(1, 2)))[0]
if len(ks) == 0:
    print("identical")
    sys.exit(0)
k0 = int(ks[0])
envs = np.nonzero(diff[k0].any(axis=1))[0]
print(f"first differing step {k0}: {len(envs)} envs: {envs[:40].tolist()}")
for e in envs[:8]:
    cols = np.nonzero(diff[k0, e])[0]
    done_prev = a[k0 - 1, e, n_obs + 1] if k0 else None
    print(f" env {e} (group {e // 64}, lane {e % 64}, vehicle {None if vid is None else int(vid[e])}) done at step {k0 - 1}: {done_prev}; columns {cols.tolist()}")
    print("   single  ", a[k0, e, cols[:8]])
    print("   resident", b[k0, e, cols[:8]])
    print("   max abs diff in row", float(np.nanmax(np.abs(a[k0, e] - b[k0, e]))))
# a later episode replaying an earlier one?  (first rows of the episodes: the start pose dominates them)
firsts = [0] + [k + 1 for k in range(K - 1) if a[k, 0, n_obs + 1] > 0.5]
print("episode starts (single launches):", firsts)
for k1 in firsts:
    if k1 >= k0 - 1 and k1 < K:
        for k2 in firsts:
            if k2 < k1:
                print(f"  resident step {k1} vs single step {k2}: median |diff| of obs[:16] {float(np.median(np.abs(b[k1, :, :16] - a[k2, :, :16]))):.3g};"
                      f"  single step {k1} vs single step {k2}: {float(np.median(np.abs(a[k1, :, :16] - a[k2, :, :16]))):.3g}")
