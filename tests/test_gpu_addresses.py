"""
The step kernel addresses its buffers as `SGPR base + 32-bit lane offset`, and the base is rebuilt from two 32-bit words
(readfirstlane).  Regression test for round 3's sign-extension bug in the observation-tile store (a low address word with
bit 31 set turned the base into 0xffff....: GPU memory fault): caller-owned buffers (actions in, packed rows out) are
placed at addresses whose LOW word has bit 31 set / clear, and right below a 4 GiB boundary so that a row range crosses
it; every placement must reproduce the rows of a reference placement bit for bit.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("config_id,n_envs", [(2, 4096), (3, 8192), (4, 4096), (5, 4096)])
def test_caller_buffers_anywhere_in_the_address_space(config_id, n_envs):
    import torch
    import bench
    from gym_dockauv_amd.envs.batched import BatchedDocking3d
    dev = torch.device("cuda", 0)
    wl = bench.workload(config_id, n_envs)
    arena = torch.empty(5 * 2 ** 30 + 2 ** 28, dtype=torch.uint8, device=dev)   # > 4 GiB: every low-word pattern occurs
    base = arena.data_ptr()

    def make_env():
        env = BatchedDocking3d(wl["cfg"], num_envs=n_envs, scenario=wl["scenario"], device=0, precision="f32",
                               reset_mode="device", device_seed=7, rng="batched", vehicles=wl["vehicles"])
        env._gen = np.random.default_rng(3)
        env.reset()
        return env

    env = make_env()
    n_obs, n_u = env.n_observations, env.n_u
    row_bytes, act_bytes = n_envs * (n_obs + 2) * 4, n_envs * n_u * 4
    env.close()
    acts = (torch.rand((3, n_envs, n_u), device=dev, generator=torch.Generator(device=dev).manual_seed(1)) * 2 - 1).contiguous()

    def placement(lo_word_target):
        """byte offset into the arena at which the address's low 32-bit word equals lo_word_target (256-B aligned)"""
        off = (lo_word_target - base) % 2 ** 32
        off = (off + 255) & ~255
        assert off + row_bytes + act_bytes + 512 < arena.numel()
        return off

    def run(off):
        env = make_env()
        out = arena[off:off + row_bytes].view(torch.float32).view(n_envs, n_obs + 2)
        a_off = (off + row_bytes + 255) & ~255
        rows = []
        for t in range(3):
            a = arena[a_off:a_off + act_bytes].view(torch.float32).view(n_envs, n_u)
            a.copy_(acts[t])
            env.step_device(a.data_ptr(), out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream, packed=True)
            torch.cuda.synchronize()
            rows.append(out.clone())
        env.synchronize()
        env.close()
        return out.data_ptr(), torch.stack(rows)

    ptr0, ref = run(placement(0x10000000))
    assert (ptr0 & 0x80000000) == 0
    for target in (0x80000000, 0xC0000000, 0xFFFFF000 - (row_bytes // 2 & ~255), 0x7FFFF000 - (row_bytes // 2 & ~255)):
        ptr, rows = run(placement(target % 2 ** 32))
        assert torch.equal(rows.view(torch.int32), ref.view(torch.int32)), f"rows differ for an output buffer at {ptr:#x}"
    del arena
